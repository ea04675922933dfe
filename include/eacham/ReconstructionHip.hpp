// ReconstructionHip.hpp — eacham's ReconstructionManager and FindBestPair on the device library.
//
//   class ReconstructionManager { MatchTwoView RecoverPoseTwoView(id1, id2, K) const; bool RecoverPosePnP(id1, id2, K); }
//       /root/reference/modules/sfm/reconstruction/ReconstructionManager.h:13-41, .cpp:47-183, :185-240
//   std::pair<unsigned, unsigned> FindBestPair(graph, map, reconstructor, K, minInitialInliers)
//       /root/reference/modules/sfm/utils/Utils.h:24-70 (call site apps/sfm/main.cpp:161-162)
//
// Same names, arguments and effects on Graph / Node / Map as the reference's; the OpenCV calls inside are the drop-ins of
// TwoViewHip.hpp / PnPHip.hpp (findEssentialMat, findHomography, decomposeHomographyMat, recoverPose, solvePnPRansac) and the
// per-match triangulation loops are eacham_two_view_points. Like ReferenceGlue.hpp the bodies are templates that only use the
// reference's accessor NAMES (modules/sfm/data/{Graph,Node,Map}.h), so they compile against eacham's own headers and, in this
// repository's tests, against tests/cpp/ref_standins.hpp (tests/cpp/sfm_loop_driver.cpp runs the whole loop of
// apps/sfm/main.cpp:150-240 on them).
// Differences a maintainer should know: the matches of a factor are visited in ascending keypoint id (the reference iterates
// an unordered_map: no order to keep); the estimators draw their samples from OpenCV's own stream by default (CvSampling.hpp:
// cv::RNG seeded (uint64)-1 per call + getSubset, restated from memory) or, with Sampling::Counter, from the seeded counter-based generator.
#pragma once

#include <algorithm>
#include <array>
#include <cstdio>
#include <limits>
#include <memory>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "PnPHip.hpp"
#include "TwoViewHip.hpp"

namespace eacham {
namespace hip {
namespace glue {

struct MatchTwoViewHip {   // MatchTwoView of modules/sfm/data/Types.h: (id2d in node 1, id2d in node 2, point3d) + the relative transform
    std::vector<std::tuple<unsigned, unsigned, std::array<double, 3>>> matches;
    std::array<double, 16> transform{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // row-major, camera 1 -> camera 2
};

template <class GraphT, class MapT>
class ReconstructionManagerHip {
public:
    // inlierThresholdPx = 0 (default) is the reference: E_Inliers / H_Inliers are the LMedS masks. Those masks are SCALE-FREE
    // (inlier = error below 2.5 * 1.4826 * sqrt(median), whatever the median is), so the best homography of a thoroughly
    // non-planar pair still calls nearly every match an inlier, H_E_ratio is ~1 for almost any pair, and the reference
    // initialises through decomposeHomographyMat, leaving it to FindBestPair's min_inliers gate to throw such pairs out.
    // inlierThresholdPx > 0 counts the inliers of the two winning models at that pixel threshold instead (what the `4.0`
    // the reference passes to both estimators would mean under RANSAC; LMEDS ignores it): an option, not the reference's behaviour.
    ReconstructionManagerHip(Context& ctx, std::shared_ptr<GraphT> graph, std::shared_ptr<MapT> map, float maxReprError, float minTriAngle,
                             int minPnpInliers, uint64_t seed = 12345, float inlierThresholdPx = 0.0f, Sampling sampling = Sampling::OpenCV)
        : ctx_(ctx), graph_(std::move(graph)), map_(std::move(map)), maxReprError_(maxReprError), minTriAngle_(minTriAngle),
          minPnpInliers_(minPnpInliers), seed_(seed), inlierThresholdPx_(inlierThresholdPx), sampling_(sampling) {}

    // ReconstructionManager.cpp:47-183
    template <class MatT>
    MatchTwoViewHip RecoverPoseTwoView(unsigned id1, unsigned id2, const MatT& K) const {
        MatchTwoViewHip result;
        auto* node1 = graph_->Get(id1);
        auto* node2 = graph_->Get(id2);
        std::vector<std::pair<unsigned, unsigned>> ms;
        for (const auto& m : node1->GetFactor(id2).matches) ms.emplace_back(m.first, m.second);
        std::sort(ms.begin(), ms.end());
        std::vector<double> pts1, pts2;
        for (const auto& m : ms) {
            const auto& a = node1->GetKeyPoint(m.first);
            const auto& b = node2->GetKeyPoint(m.second);
            pts1.push_back(a.x), pts1.push_back(a.y), pts2.push_back(b.x), pts2.push_back(b.y);
        }
        const double K9[9] = {K.template at<double>(0, 0), 0, K.template at<double>(0, 2), 0, K.template at<double>(1, 1),
                              K.template at<double>(1, 2), 0, 0, 1};
        // :57-61 findEssentialMat(pts1, pts2, focal = K(0,0), pp, LMEDS, 0.99, 4.0, 1000, mask)
        const double K4[4] = {K9[0], K9[0], K9[2], K9[5]};
        const uint64_t s = seed_ + 0x9E3779B97F4A7C15ull * ((uint64_t)id1 * 65536 + id2);
        const RobustModel E = FindEssentialMat(ctx_, pts1, pts2, K4, 1000, s, 0.99, sampling_);
        // :75 findHomography(pts1, pts2, LMEDS, 4.0, mask2, 100, 0.999)
        const RobustModel H = FindHomography(ctx_, pts1, pts2, 100, s + 1, 0.999, sampling_);
        if (!E.ok) return result;
        int eInliers = E.inliers, hInliers = H.ok ? H.inliers : 0;
        if (inlierThresholdPx_ > 0.0f) {
            const int n = (int)ms.size();
            int32_t c = 0;
            const float te = (inlierThresholdPx_ / (float)K9[0]) * (inlierThresholdPx_ / (float)K9[0]);   // Sampson error lives in normalised coordinates
            ctx_.check(eacham_score_hypotheses(ctx_.get(), EACHAM_SCORE_ESSENTIAL, n, pts1.data(), pts2.data(), 1, E.model.data(), K4, te, nullptr, &c, nullptr));
            eInliers = c;
            if (H.ok) {
                ctx_.check(eacham_score_hypotheses(ctx_.get(), EACHAM_SCORE_HOMOGRAPHY, n, pts1.data(), pts2.data(), 1, H.model.data(), nullptr,
                                                  inlierThresholdPx_ * inlierThresholdPx_, nullptr, &c, nullptr));
                hInliers = c;
            }
        }
        const float ratio = hInliers > 0 && eInliers > 0 ? (float)hInliers / (float)eInliers : 0.0f;   // :87
#ifdef EACHAM_RECON_DEBUG
        std::fprintf(stderr, "two-view %u-%u: %zu matches, E inliers %d (median %g), H inliers %d (median %g), ratio %.3f\n", id1, id2, ms.size(), eInliers,
                     (double)E.median, hInliers, (double)H.median, (double)ratio);
#endif
        if (ratio > 0.9f) {                                                                        // :89-150
            const auto sols = DecomposeHomographyMat(H.model, K9);
            std::vector<double> T;
            for (const auto& m : sols) {
                const double M[16] = {m.R[0], m.R[1], m.R[2], m.t[0], m.R[3], m.R[4], m.R[5], m.t[1], m.R[6], m.R[7], m.R[8], m.t[2], 0, 0, 0, 1};
                T.insert(T.end(), M, M + 16);
            }
            if (sols.empty()) return result;
            const auto tv = TwoViewPoints(ctx_, pts1, pts2, K9, T, maxReprError_, minTriAngle_, true);
            const int best = BestTwoViewSolution(tv);
            if (best >= 0) {
                for (const auto& m : tv[best].matches) result.matches.emplace_back(ms[m.first].first, ms[m.first].second, m.second);
                std::copy(&T[16 * best], &T[16 * best] + 16, result.transform.begin());
            }
        } else {                                                                                   // :152-180
            const RecoveredPose p = RecoverPose(ctx_, E.model, pts1, pts2, K9, 50.0, &E.mask);
            const auto M = p.transform();
            const std::vector<double> T(M.begin(), M.end());
            const auto tv = TwoViewPoints(ctx_, pts1, pts2, K9, T, maxReprError_, minTriAngle_, false);
            for (const auto& m : tv[0].matches) result.matches.emplace_back(ms[m.first].first, ms[m.first].second, m.second);
            result.transform = M;
        }
        return result;
    }

    // ReconstructionManager.cpp:185-240
    template <class MatT>
    bool RecoverPosePnP(unsigned id1, unsigned id2, const MatT& K) {
        auto* node1 = graph_->Get(id1);
        auto* node2 = graph_->Get(id2);
        auto& factor = node1->GetFactor(id2);
        std::vector<std::pair<unsigned, unsigned>> ms;
        for (const auto& m : factor.matches) ms.emplace_back(m.first, m.second);
        std::sort(ms.begin(), ms.end());
        std::vector<double> pts3d, pts2d;
        for (const auto& m : ms)
            if (node1->HasPoint3d(m.first)) {
                const auto X = map_->Get(node1->GetPoint3d(m.first));
                pts3d.push_back(X(0)), pts3d.push_back(X(1)), pts3d.push_back(X(2));
                const auto& b = node2->GetKeyPoint(m.second);
                pts2d.push_back(b.x), pts2d.push_back(b.y);
            }
        if ((int)(pts2d.size() / 2) < minPnpInliers_) return false;                               // :214-217
        const double K9[9] = {K.template at<double>(0, 0), 0, K.template at<double>(0, 2), 0, K.template at<double>(1, 1),
                              K.template at<double>(1, 2), 0, 0, 1};
        const PnPResult r = SolvePnPRansac(ctx_, pts3d, pts2d, K9, 10000, 4.0f, 0.999, seed_ + 0xD1B54A32D192ED03ull * ((uint64_t)id1 * 65536 + id2), sampling_);
        if (!r.ok || (int)r.inliers.size() < minPnpInliers_) return false;                         // :229-234
        using Mat4 = std::decay_t<decltype(node2->GetTransform())>;
        Mat4 M;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) M(i, j) = r.R[3 * i + j];
            M(i, 3) = r.t[i];
            M(3, i) = 0.0;
        }
        M(3, 3) = 1.0;
        factor.transform = M;                                                                      // :236-240
        node2->SetTransform(M);
        node2->SetValid(true);
        return true;
    }

    Context& context() const { return ctx_; }

private:
    Context& ctx_;
    std::shared_ptr<GraphT> graph_;
    std::shared_ptr<MapT> map_;
    float maxReprError_, minTriAngle_;
    int minPnpInliers_;
    uint64_t seed_;
    float inlierThresholdPx_;
    Sampling sampling_;   // OpenCV's own sample stream (default) or the counter-based generator seeded with seed_
};

// utils::FindBestPair (Utils.h:24-70): the first pair of connected nodes whose two-view reconstruction passes in BOTH directions;
// node 1 becomes the fixed origin, node 2 gets the relative transform, the kept matches become two-view map points.
template <class GraphT, class MapT, class MatT>
inline std::pair<unsigned, unsigned> FindBestPair(const std::shared_ptr<GraphT>& graph, const std::shared_ptr<MapT>& map,
                                                  const ReconstructionManagerHip<GraphT, MapT>& reconstructor, const MatT& K,
                                                  unsigned minInitialInliers) {
    for (const auto& entry : graph->GetNodes()) {
        const unsigned id1 = entry.first;
        auto* node1 = entry.second;
        std::vector<unsigned> neighbours;
        for (const auto& f : node1->GetFactors()) neighbours.push_back(f.first);
        std::sort(neighbours.begin(), neighbours.end());   // (the reference walks an unordered_map)
        for (const unsigned id2 : neighbours) {
            const MatchTwoViewHip rec1 = reconstructor.RecoverPoseTwoView(id1, id2, K);
            const MatchTwoViewHip rec2 = reconstructor.RecoverPoseTwoView(id2, id1, K);
            if (rec1.matches.size() > minInitialInliers && rec2.matches.size() > minInitialInliers) {
                graph->FixNode(id1);
                node1->GetFactor(id2).quality = (unsigned)rec1.matches.size();
                auto* node2 = graph->Get(id2);
                using Mat4 = std::decay_t<decltype(node1->GetTransform())>;
                Mat4 I, M;
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) I(i, j) = i == j ? 1.0 : 0.0, M(i, j) = rec1.transform[4 * i + j];
                node1->SetTransform(I);
                node1->SetValid(true);
                node2->SetTransform(M);
                node2->SetValid(true);
                using Vec3 = std::decay_t<decltype(map->Get(0u))>;
                for (const auto& m : rec1.matches) {
                    Vec3 X, colour;
                    for (int e = 0; e < 3; ++e) X(e) = std::get<2>(m)[e], colour(e) = 0.3;
                    const unsigned id3d = map->Add(X, colour);
                    node1->SetPoint3d(std::get<0>(m), id3d, true);
                    node2->SetPoint3d(std::get<1>(m), id3d, true);
                    map->AddObserver(id1, std::get<0>(m), id3d);
                    map->AddObserver(id2, std::get<1>(m), id3d);
                }
                return {id1, id2};
            }
        }
    }
    return {std::numeric_limits<unsigned>::max(), std::numeric_limits<unsigned>::max()};
}

}  // namespace glue
}  // namespace hip
}  // namespace eacham
