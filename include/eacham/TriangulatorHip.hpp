// TriangulatorHip.hpp — C++ adapter that restates eacham's triangulation calls on top of the C-ABI.
//
//   bool TriangulatePointRansac(const std::vector<EstimatorData>& data, Eigen::Vector3d& point3d,
//                               std::vector<bool>& inliers, const float maxReprError, const float minTriAngle);
//       /root/reference/modules/sfm/reconstruction/Triangulator.h:18-24,37-39, .cpp:96-186
//   void TriangulateFrame(const unsigned frameId, std::shared_ptr<graph_t> graph, std::shared_ptr<Map> map,
//                         const cv::Mat& K, const unsigned minObservers, const float maxReprError,
//                         const float minTriAngle);
//       Triangulator.h:41-43, .cpp:188-300; call sites apps/sfm/main.cpp:203-210
//
// OpenCV / Eigen are not required: graph and map are seen through small views a caller fills from
// eacham's Graph/Node/Map accessors. TriangulateFrame here performs the reference's
// graph walk on the host and hands the arithmetic to the device in two batches:
//   1. the re-observation gate (CalcReprojectionError of existing map points, :222-236)
//      -> eacham_reprojection_errors, then the observer bookkeeping in the reference's order;
//   2. every candidate track with >= minObservers observers (:248-283) -> eacham_triangulate_tracks,
//      then Map::Add / AddObserver / RemoveObserver / UpdateStatus for accepted tracks.
// The reference iterates `factors` in unordered_map order (Node.h:223), i.e. in no defined order;
// this adapter visits neighbours in ascending id and matches in the order given.
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <utility>
#include <vector>

#include "../eacham_hip.h"
#include "FeatureMatcherHip.hpp"  // Context
#include "FlatMap.hpp"

namespace eacham {
namespace hip {

struct EstimatorData {  // Triangulator.h:18-24 (K is passed once per call instead of per observation)
    double transform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // row-major world->camera
    double point2d[2] = {0, 0};
};

// K: 3x3 row-major camera matrix; entries (0,0) (1,1) (0,2) (1,2) are read.
inline bool TriangulatePointRansac(Context& ctx, const std::vector<EstimatorData>& data, const double* K, double* point3d,
                                   std::vector<bool>& inliers, const float maxReprError, const float minTriAngle) {
    const int m = (int)data.size();
    std::vector<double> T((size_t)m * 16), uv((size_t)m * 2);
    std::vector<uint32_t> frame(m);
    for (int i = 0; i < m; ++i) {
        for (int k = 0; k < 16; ++k) T[(size_t)i * 16 + k] = data[i].transform[k];
        uv[2 * i] = data[i].point2d[0];
        uv[2 * i + 1] = data[i].point2d[1];
        frame[i] = (uint32_t)i;
    }
    const int32_t track_ptr[2] = {0, m};
    const double K4[4] = {K[0], K[4], K[2], K[5]};
    int32_t status = 0;
    std::vector<uint8_t> mask(m > 0 ? m : 1);
    ctx.check(eacham_triangulate_tracks(ctx.get(), T.data(), m, 1, track_ptr, frame.data(), uv.data(), K4, maxReprError,
                                        minTriAngle, point3d, &status, mask.data()));
    inliers.clear();
    // the reference leaves `inliers` empty when it returns before scoring (:104-110) or when no
    // pair passes the angle gate (:176-184 copies an empty best mask)
    bool any = false;
    for (int i = 0; i < m; ++i) any = any || mask[i];
    if (any)
        for (int i = 0; i < m; ++i) inliers.push_back(mask[i] != 0);
    return (status & 1) != 0;
}

// ---- two-view structure for candidate relative poses --------------------------------------------
// The per-match loops of ReconstructionManager::RecoverPoseTwoView
// (/root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:118-143 over the solutions of
// cv::decomposeHomographyMat, :162-186 for the pose of cv::recoverPose): triangulate every match
// against each candidate transform, keep it iff z > 0, the reprojection error in camera 1 is below
// maxReprError and the triangulation angle passes (strictly greater in the homography branch).
struct TwoViewSolution {
    std::vector<std::pair<unsigned, std::array<double, 3>>> matches;  // (index into the match list, point3d)
};

inline std::vector<TwoViewSolution> TwoViewPoints(Context& ctx, const std::vector<double>& uv1, const std::vector<double>& uv2,
                                                  const double* K, const std::vector<double>& transforms /* k x 16 */,
                                                  float maxReprError, float minTriAngle, bool homographyBranch) {
    const int n = (int)(uv1.size() / 2), nt = (int)(transforms.size() / 16);
    if (uv2.size() != uv1.size()) throw std::runtime_error("TwoViewPoints: point lists disagree");
    const double K4[4] = {K[0], K[4], K[2], K[5]};
    std::vector<double> pts((size_t)3 * n * nt + 3);
    std::vector<uint8_t> keep((size_t)n * nt + 1);
    std::vector<int32_t> counts(nt + 1);
    ctx.check(eacham_two_view_points(ctx.get(), n, uv1.data(), uv2.data(), K4, nt, transforms.data(), maxReprError, minTriAngle,
                                     homographyBranch ? 1 : 0, pts.data(), keep.data(), counts.data()));
    std::vector<TwoViewSolution> out(nt);
    for (int k = 0; k < nt; ++k)
        for (int i = 0; i < n; ++i)
            if (keep[(size_t)k * n + i]) {
                const double* p = &pts[3 * ((size_t)k * n + i)];
                out[k].matches.push_back({(unsigned)i, {p[0], p[1], p[2]}});
            }
    return out;
}

// The reference's choice among the homography solutions (:139-150): the first with the strictly
// largest number of kept matches, accepted only above 20 matches; -1 otherwise.
inline int BestTwoViewSolution(const std::vector<TwoViewSolution>& sols) {
    int best = -1;
    size_t bestSize = 0;
    for (size_t k = 0; k < sols.size(); ++k)
        if (sols[k].matches.size() > bestSize) {
            bestSize = sols[k].matches.size();
            best = (int)k;
        }
    return bestSize > 20 ? best : -1;
}

// ---- hypothesis scoring for the robust estimators (ReconstructionManager.cpp:57-61, :75, :227-228) -------------
// One call scores every candidate model against every correspondence: inlier counts under `threshold` (RANSAC,
// err <= threshold with the SQUARED pixel threshold, as OpenCV's findInliers compares) and medians (LMedS).
//   kind EACHAM_SCORE_ESSENTIAL : a, b = x0 y0 x1 y1 ... pixels of view 1 / 2, models = 9 doubles per E, K4 = fx fy cx cy
//                                 (nullptr: the points are already normalised)
//   kind EACHAM_SCORE_HOMOGRAPHY: the same points, models = 9 doubles per H (H[8] taken as 1), K4 unused
//   kind EACHAM_SCORE_PNP       : a = X0 Y0 Z0 ... object points, b = pixels, models = 12 doubles per pose (R row-major | t)
// The caller keeps OpenCV's sampling + minimal solvers (cv::findEssentialMat's five-point, EPnP) and hands over the
// candidates; `best_by_inliers()` / `best_by_median()` give the RANSAC / LMedS choice (first best on ties).
struct HypothesisScores {
    std::vector<int32_t> inliers;
    std::vector<float> medians;
    int best_by_inliers() const {
        int best = -1;
        for (size_t m = 0; m < inliers.size(); ++m)
            if (best < 0 || inliers[m] > inliers[best]) best = (int)m;
        return best;
    }
    int best_by_median() const {
        int best = -1;
        for (size_t m = 0; m < medians.size(); ++m)
            if (medians[m] == medians[m] && (best < 0 || medians[m] < medians[best])) best = (int)m;
        return best;
    }
};

inline HypothesisScores ScoreHypotheses(Context& ctx, int kind, const std::vector<double>& a, const std::vector<double>& b,
                                        const std::vector<double>& models, const double* K4, float threshold) {
    const int stride_a = kind == EACHAM_SCORE_PNP ? 3 : 2, stride_m = kind == EACHAM_SCORE_PNP ? 12 : 9;
    const int n = (int)(b.size() / 2), nm = (int)(models.size() / stride_m);
    if ((size_t)n * stride_a != a.size() || (size_t)n * 2 != b.size() || (size_t)nm * stride_m != models.size())
        throw std::runtime_error("ScoreHypotheses: array sizes disagree");
    HypothesisScores out;
    out.inliers.resize(nm);
    out.medians.resize(nm);
    ctx.check(eacham_score_hypotheses(ctx.get(), kind, n, a.data(), b.data(), nm, models.data(), K4, threshold, nullptr,
                                      out.inliers.data(), out.medians.data()));
    return out;
}

// ---- views for TriangulateFrame ---------------------------------------------------------------

struct TriNodeView {
    bool valid = false;                                   // Node::IsValid()
    double transform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::vector<float> keypoints;                         // x0 y0 x1 y1 ... (cv::Point2f)
    FlatMap points3d;                                     // keypoint -> map point (Node::SetPoint3d / HasPoint3d)
    std::map<unsigned, std::vector<std::pair<unsigned, unsigned>>> factors;  // neighbour id -> (m1, m2) matches
};
struct TriGraphView {
    std::map<unsigned, TriNodeView> nodes;
};
struct TriMapPoint {
    double point3d[3] = {0, 0, 0};
    bool isValid = false;
    FlatMap observers;                                    // frame -> keypoint (MapPointData::observers)
};
struct TriMapView {                                       // the operations of modules/sfm/data/Map.h the walk uses
    std::map<unsigned, TriMapPoint> points;
    unsigned mapPointId = 0;
    unsigned Add(const double* p) {
        ++mapPointId;
        TriMapPoint mp;
        mp.point3d[0] = p[0]; mp.point3d[1] = p[1]; mp.point3d[2] = p[2];
        points[mapPointId] = mp;
        return mapPointId;
    }
    TriMapPoint& at(unsigned id) {
        auto it = points.find(id);
        if (it == points.end()) throw std::runtime_error("Map: point is not found");
        return it->second;
    }
};

struct TriangulateFrameReport {
    unsigned total = 0, added = 0, reobserved = 0;
};

inline TriangulateFrameReport TriangulateFrame(Context& ctx, const unsigned frameId, TriGraphView& graph, TriMapView& map,
                                               const double* K, const unsigned minObservers, const float maxReprError,
                                               const float minTriAngle) {
    auto node_at = [&](unsigned id) -> TriNodeView& {
        auto it = graph.nodes.find(id);
        if (it == graph.nodes.end()) throw std::runtime_error("Node is null");
        return it->second;
    };
    auto keypoint = [](const TriNodeView& n, unsigned k, double* out) {
        if (2 * (size_t)k + 1 >= n.keypoints.size()) throw std::runtime_error("TriangulateFrame: keypoint out of range");
        out[0] = n.keypoints[2 * k];
        out[1] = n.keypoints[2 * k + 1];
    };
    TriNodeView& current = node_at(frameId);
    const double K4[4] = {K[0], K[4], K[2], K[5]};
    TriangulateFrameReport rep;

    // ---- phase 1: matches whose partner already has a map point (:208-240) ----
    struct Cand { unsigned id, m1, m2, point3d2; };
    std::vector<Cand> cands;
    std::vector<uint32_t> cframe;
    std::vector<double> cpts, cuv;
    for (const auto& f : current.factors) {
        const TriNodeView& other = node_at(f.first);
        if (!other.valid) continue;
        for (const auto& mm : f.second) {
            const auto has = other.points3d.find(mm.second);
            if (has == other.points3d.end()) continue;
            const TriMapPoint& mp = map.at(has->second);
            cands.push_back({f.first, mm.first, mm.second, has->second});
            cframe.push_back(0);
            cpts.insert(cpts.end(), mp.point3d, mp.point3d + 3);
            double p[2];
            keypoint(current, mm.first, p);
            cuv.push_back(p[0]);
            cuv.push_back(p[1]);
        }
    }
    std::vector<float> cerr(cands.size());
    ctx.check(eacham_reprojection_errors(ctx.get(), current.transform, 1, (int)cands.size(), cframe.data(), cpts.data(),
                                         cuv.data(), K4, cerr.data()));

    std::map<unsigned, FlatMap> observersFull;
    size_t ci = 0;
    for (const auto& f : current.factors) {
        const TriNodeView& other = node_at(f.first);
        if (!other.valid) continue;
        for (const auto& mm : f.second) {
            const auto has = other.points3d.find(mm.second);
            if (has != other.points3d.end()) {
                const float err = cerr[ci++];
                if (map.at(has->second).observers.size() > 2 && err < maxReprError) {  // observer count read in walk order
                    current.points3d[mm.first] = has->second;                          // SetPoint3d(m1, point3d2, false)
                    map.at(has->second).observers[frameId] = mm.first;                 // AddObserver
                    ++rep.reobserved;
                    continue;
                }
            }
            observersFull[mm.first][frameId] = mm.first;
            observersFull[mm.first][f.first] = mm.second;
        }
    }

    // ---- phase 2: candidate tracks (:248-262) ----
    std::map<unsigned, uint32_t> frameRow;   // node id -> row of the transform table
    std::vector<double> transforms, uv;
    std::vector<int32_t> trackPtr{0};
    std::vector<uint32_t> obsFrame;
    std::vector<const FlatMap*> trackObs;
    for (const auto& kv : observersFull) {
        if (kv.second.size() < minObservers) continue;
        for (const auto& ob : kv.second) {
            const TriNodeView& n = node_at(ob.first);
            auto ins = frameRow.insert({ob.first, (uint32_t)frameRow.size()});
            if (ins.second) transforms.insert(transforms.end(), n.transform, n.transform + 16);
            obsFrame.push_back(ins.first->second);
            double p[2];
            keypoint(n, ob.second, p);
            uv.push_back(p[0]);
            uv.push_back(p[1]);
        }
        trackPtr.push_back((int32_t)obsFrame.size());
        trackObs.push_back(&kv.second);
    }
    const int nTracks = (int)trackObs.size();
    std::vector<double> pts((size_t)nTracks * 3 + 3);
    std::vector<int32_t> status(nTracks + 1);
    std::vector<uint8_t> masks(obsFrame.size() + 1);
    ctx.check(eacham_triangulate_tracks(ctx.get(), transforms.data(), (int)frameRow.size(), nTracks, trackPtr.data(),
                                        obsFrame.data(), uv.data(), K4, maxReprError, minTriAngle, pts.data(), status.data(),
                                        masks.data()));

    // ---- map bookkeeping for accepted tracks (:270-296) ----
    for (int t = 0; t < nTracks; ++t) {
        if (status[t] == 3) {
            const unsigned mapPointId = map.Add(&pts[(size_t)t * 3]);
            for (const auto& ob : *trackObs[t]) {
                TriNodeView& n = node_at(ob.first);
                const auto old = n.points3d.find(ob.second);
                if (old != n.points3d.end()) {
                    TriMapPoint& op = map.at(old->second);
                    op.observers.erase(ob.first);   // RemoveObserver
                    op.isValid = false;             // UpdateStatus(old, false)
                }
                n.points3d[ob.second] = mapPointId;
                map.at(mapPointId).observers[ob.first] = ob.second;
            }
            map.at(mapPointId).isValid = true;
            ++rep.added;
        }
        ++rep.total;
    }
    return rep;
}

}  // namespace hip
}  // namespace eacham
