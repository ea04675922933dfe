"""eacham_amd — MI355X-native matching + bundle-adjustment hot path of fatlipp/eacham.

The product is the HIP library behind include/eacham_hip.h (eacham_amd/csrc). This package is the
thin host-side mirror of the reference interfaces used by tests and bench.py.
"""
from .capi import EachamError  # noqa: F401
from .matcher import FeatureMatcherHip, HipContext, MIN_DIRECTED, MIN_MUTUAL, RATIO  # noqa: F401
