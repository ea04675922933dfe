// FeatureMatcherHip.hpp — C++ adapter that restates eacham's matcher interface on top of the C-ABI.
//
//   IFeatureMatcher<T>::Match(const T&, const T&) -> std::unordered_map<unsigned, unsigned>
//       /root/reference/modules/base/features/IFeatureMatcher.h:8-20
//   FeatureMatcherFlann(const float inliersRatio); Match(const cv::Mat&, const cv::Mat&)
//       /root/reference/modules/base/features/FeatureMatcherFlann.h:11-25, .cpp:8-30
//
// Header-only; link against libeacham_hip.so. Errors become std::runtime_error like the reference's
// own accessors (Node.h:78-98); the C-ABI itself never throws. Thread-safe: one shared instance may
// be called from many threads (apps/sfm/main.cpp:98-109) — calls serialise on the context.
#pragma once

#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../eacham_hip.h"

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define EACHAM_HIP_HAVE_OPENCV 1
#endif
#endif

namespace eacham {
namespace hip {

// N x D row-major fp32 descriptors: the memory layout of the cv::Mat FeatureExtractorSift returns.
struct DescriptorView {
    const float* data = nullptr;
    int rows = 0;
    int dim = 0;
};

// Same shape as eacham::IFeatureMatcher<T>.
template <typename T>
class IFeatureMatcher {
public:
    using MatchType = std::unordered_map<unsigned, unsigned>;
    virtual ~IFeatureMatcher() = default;
    virtual MatchType Match(const T& descriptor1, const T& descriptor2) = 0;
};

class Context {
public:
    explicit Context(int device = 0) {
        if (eacham_ctx_create(device, &ctx_) != EACHAM_OK)
            throw std::runtime_error("eacham_hip: no usable HIP device (the hot path has no CPU fallback)");
    }
    ~Context() { eacham_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    eacham_ctx* get() const { return ctx_; }
    void check(int rc) const {
        if (rc != EACHAM_OK) throw std::runtime_error(std::string("eacham_hip: ") + eacham_last_error(ctx_));
    }

private:
    eacham_ctx* ctx_ = nullptr;
};

class FeatureMatcherHip : public IFeatureMatcher<DescriptorView> {
public:
    // `inliersRatio` is kept for signature parity; as in the reference it is NOT the ratio of the
    // test, which is the literal 0.8 (FeatureMatcherFlann.cpp:23) unless `ratio` overrides it.
    explicit FeatureMatcherHip(float inliersRatio = 0.8f, int device = 0, double ratio = 0.8)
        : inliersRatio_(inliersRatio), ratio_(ratio), ctx_(device) {}

    MatchType Match(const DescriptorView& d1, const DescriptorView& d2) override {
        std::lock_guard<std::mutex> lock(mu_);  // the two scratch frame slots are shared
        // SIFT-style integer descriptors take the exact int8 path; anything else (SuperPoint /
        // LightGlue floats, modules/onnx/lightglue/feature/Types.h:11-14) the fp32 MFMA path.
        ctx_.check(eacham_clear_descriptors(ctx_.get()));
        int rc = eacham_upload_descriptors(ctx_.get(), kSlotA, d1.data, d1.rows, d1.dim);
        if (rc == EACHAM_OK) rc = eacham_upload_descriptors(ctx_.get(), kSlotB, d2.data, d2.rows, d2.dim);
        if (rc == EACHAM_ERR_NOT_INTEGER || rc == EACHAM_ERR_UNSUPPORTED) {
            ctx_.check(eacham_clear_descriptors(ctx_.get()));
            ctx_.check(eacham_upload_descriptors_f32(ctx_.get(), kSlotA, d1.data, d1.rows, d1.dim));
            ctx_.check(eacham_upload_descriptors_f32(ctx_.get(), kSlotB, d2.data, d2.rows, d2.dim));
        } else {
            ctx_.check(rc);
        }
        std::vector<uint32_t> q(d1.rows > 0 ? d1.rows : 1), t(q.size());
        int count = 0;
        ctx_.check(eacham_match_pair(ctx_.get(), kSlotA, kSlotB, ratio_, q.data(), t.data(), (int)q.size(), &count));
        MatchType out;
        out.reserve(count);
        for (int k = 0; k < count; ++k) out.insert({q[k], t[k]});
        return out;
    }

#ifdef EACHAM_HIP_HAVE_OPENCV
    // Drop-in for FeatureMatcherFlann::Match(const cv::Mat&, const cv::Mat&): CV_32F, continuous.
    MatchType Match(const cv::Mat& d1, const cv::Mat& d2) {
        if (d1.type() != CV_32F || d2.type() != CV_32F || !d1.isContinuous() || !d2.isContinuous())
            throw std::runtime_error("eacham_hip: descriptors must be continuous CV_32F matrices");
        return Match(DescriptorView{d1.ptr<float>(), d1.rows, d1.cols}, DescriptorView{d2.ptr<float>(), d2.rows, d2.cols});
    }
#endif

    Context& context() { return ctx_; }

private:
    static constexpr int kSlotA = 65534, kSlotB = 65535;
    float inliersRatio_;
    double ratio_;
    Context ctx_;
    std::mutex mu_;
};

// The pair loop of apps/sfm/main.cpp:84-147 as one call: frames are uploaded once, every unordered
// pair is matched in both directions with the mutual check and the literal thresholds 30 / 30.
struct MatchGraph {
    std::vector<int32_t> counts;   // per pair: |mutual| if the pair became an edge, else 0
    std::vector<int64_t> offsets;  // CSR
    std::vector<uint32_t> q, t;    // Graph::Connect(n1, n2, {q -> t}); the reverse edge is the inverse map
};

inline MatchGraph MatchAllPairs(Context& ctx, const std::vector<DescriptorView>& frames,
                                const std::vector<std::pair<unsigned, unsigned>>& pairs, double ratio = 0.8,
                                int min_directed = 30, int min_mutual = 30) {
    for (size_t f = 0; f < frames.size(); ++f)
        ctx.check(eacham_upload_descriptors(ctx.get(), (int)f, frames[f].data, frames[f].rows, frames[f].dim));
    std::vector<int32_t> flat(2 * pairs.size());
    int64_t cap = 0;
    for (size_t p = 0; p < pairs.size(); ++p) {
        flat[2 * p] = (int32_t)pairs[p].first;
        flat[2 * p + 1] = (int32_t)pairs[p].second;
        cap += frames.at(pairs[p].first).rows;
    }
    MatchGraph g;
    g.counts.resize(pairs.size());
    g.offsets.resize(pairs.size() + 1);
    g.q.resize(cap > 0 ? cap : 1);
    g.t.resize(g.q.size());
    int64_t total = 0;
    ctx.check(eacham_match_all_pairs(ctx.get(), flat.data(), (int)pairs.size(), ratio, min_directed, min_mutual,
                                     g.counts.data(), g.offsets.data(), g.q.data(), g.t.data(), cap, &total, nullptr));
    g.q.resize(total);
    g.t.resize(total);
    return g;
}

// ---- view-graph query on the CSR match graph ---------------------------------------------------
// std::tuple<unsigned, unsigned, unsigned> Graph::GetBestPairForValid(const std::set<unsigned>& excluded)
//     /root/reference/modules/sfm/data/Graph.h:59-106
// evaluated on the wire format MatchAllPairs returns, no Graph::Connect round trip needed.
//   valid[f]     Node::IsValid();   excluded (may be empty) = the `excluded` set as flags per frame
//   has3d[f][k]  node f HasPoint3d(k) && !IsPoint3dTwoView(k)
struct BestPair {
    unsigned id = 0xffffffffu, id2 = 0xffffffffu, points3dCount = 0;
};

inline BestPair GetBestPairForValid(Context& ctx, const std::vector<std::pair<unsigned, unsigned>>& pairs, const MatchGraph& g,
                                    const std::vector<uint8_t>& valid, const std::vector<std::vector<uint8_t>>& has3d,
                                    const std::vector<uint8_t>& excluded = {}) {
    const int nFrames = (int)valid.size();
    if ((int)has3d.size() != nFrames || (!excluded.empty() && (int)excluded.size() != nFrames))
        throw std::runtime_error("GetBestPairForValid: per-frame arrays disagree");
    std::vector<int32_t> flat(2 * pairs.size());
    for (size_t p = 0; p < pairs.size(); ++p) {
        flat[2 * p] = (int32_t)pairs[p].first;
        flat[2 * p + 1] = (int32_t)pairs[p].second;
    }
    std::vector<int64_t> kpOffsets(nFrames + 1, 0);
    for (int f = 0; f < nFrames; ++f) kpOffsets[f + 1] = kpOffsets[f] + (int64_t)has3d[f].size();
    std::vector<uint8_t> flags((size_t)kpOffsets[nFrames] + 1);
    for (int f = 0; f < nFrames; ++f) std::copy(has3d[f].begin(), has3d[f].end(), flags.begin() + kpOffsets[f]);
    uint32_t best[3] = {0, 0, 0};
    ctx.check(eacham_graph_best_pair(ctx.get(), nFrames, flat.data(), (int)pairs.size(), g.counts.data(), g.offsets.data(),
                                     g.q.data(), g.t.data(), valid.data(), excluded.empty() ? nullptr : excluded.data(),
                                     kpOffsets.data(), flags.data(), nullptr, best));
    BestPair r;
    r.id = best[0];
    r.id2 = best[1];
    r.points3dCount = best[2];
    return r;
}

}  // namespace hip
}  // namespace eacham
