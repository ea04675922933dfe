// Test driver for the C++ adapters (include/eacham/*.hpp): reads a binary fixture written by
// tests/test_cpp_adapters.py, runs FeatureMatcherHip::Match / MatchAllPairs / RefineBA through the
// C-ABI and writes the results back for comparison with the oracle.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <thread>

#ifdef EACHAM_TEST_GLUE  // the RefineBA leg goes through the reference-typed glue on stand-ins of Graph / Node / Map / cv::Mat
#include "ref_standins.hpp"
#include "eacham/ReferenceGlue.hpp"
// the drop-in definition has exactly the reference's signature (BundleAdjuster.h:13-17)
static void (*const kRefineBA)(const int, std::shared_ptr<eacham::graph_t>, std::shared_ptr<eacham::Map>, cv::Mat&,
                               const eacham::OptimizerConfig&) = &eacham::RefineBA;
#endif
#include "eacham/BundleAdjusterHip.hpp"
#include "eacham/FeatureMatcherHip.hpp"

using namespace eacham::hip;

template <class T> static std::vector<T> rd(std::ifstream& f, size_t n) {
    std::vector<T> v(n);
    f.read((char*)v.data(), sizeof(T) * n);
    return v;
}
template <class T> static T rd1(std::ifstream& f) { T v; f.read((char*)&v, sizeof(T)); return v; }
template <class T> static void wr(std::ofstream& f, const std::vector<T>& v) {
    int64_t n = (int64_t)v.size();
    f.write((char*)&n, sizeof(n));
    f.write((const char*)v.data(), sizeof(T) * v.size());
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    std::ofstream out(argv[2], std::ios::binary);
    // ---- matcher: F frames of n x dim ----
    const int F = rd1<int32_t>(in), dim = rd1<int32_t>(in);
    std::vector<std::vector<float>> store;
    std::vector<DescriptorView> frames;
    for (int f = 0; f < F; ++f) {
        const int n = rd1<int32_t>(in);
        store.push_back(rd<float>(in, (size_t)n * dim));
        frames.push_back(DescriptorView{store.back().data(), n, dim});
    }
    FeatureMatcherHip matcher(0.8f);
    IFeatureMatcher<DescriptorView>& iface = matcher;  // used through the reference-shaped interface
    // the reference calls Match() concurrently on one shared instance (apps/sfm/main.cpp:98-109)
    std::vector<IFeatureMatcher<DescriptorView>::MatchType> res(4);
    std::vector<std::thread> th;
    const int jobs[4][2] = {{0, 1}, {1, 0}, {0, 2}, {2, 1}};
    for (int k = 0; k < 4; ++k) th.emplace_back([&, k] { res[k] = iface.Match(frames[jobs[k][0]], frames[jobs[k][1]]); });
    for (auto& t : th) t.join();
    for (int k = 0; k < 4; ++k) {
        std::vector<uint32_t> flat;
        for (unsigned q = 0; q < (unsigned)frames[jobs[k][0]].rows; ++q) {
            auto it = res[k].find(q);
            if (it != res[k].end()) { flat.push_back(q); flat.push_back(it->second); }
        }
        wr(out, flat);
    }
    std::vector<std::pair<unsigned, unsigned>> pairs;
    for (unsigned i = 0; i < (unsigned)F; ++i)
        for (unsigned j = i + 1; j < (unsigned)F; ++j) pairs.push_back({i, j});
    MatchGraph g = MatchAllPairs(matcher.context(), frames, pairs, 0.8, 5, 5);
    wr(out, g.counts); wr(out, g.q); wr(out, g.t);
    {   // GetBestPairForValid on that graph: frames 0 and 1 valid, every third keypoint has a 3-D point
        std::vector<uint8_t> valid(F, 0);
        valid[0] = 1;
        if (F > 1) valid[1] = 1;
        std::vector<std::vector<uint8_t>> has3d(F);
        for (int f = 0; f < F; ++f) {
            has3d[f].resize(frames[f].rows);
            for (int k = 0; k < frames[f].rows; ++k) has3d[f][k] = valid[f] && (k % 3 == 0);
        }
        BestPair bp = GetBestPairForValid(matcher.context(), pairs, g, valid, has3d);
        std::vector<uint32_t> b = {bp.id, bp.id2, bp.points3dCount};
        wr(out, b);
    }

    // ---- RefineBA on a graph/map view ----
    GraphView graph; MapView map;
    const int nNodes = rd1<int32_t>(in);
    for (int i = 0; i < nNodes; ++i) {
        NodeView nd;
        nd.id = rd1<uint32_t>(in); nd.valid = rd1<int32_t>(in) != 0; nd.fixed = rd1<int32_t>(in) != 0;
        auto T = rd<double>(in, 16);
        std::copy(T.begin(), T.end(), nd.transform);
        const int nk = rd1<int32_t>(in);
        nd.keypoints = rd<float>(in, 2 * (size_t)nk);
        const int np = rd1<int32_t>(in);
        for (int k = 0; k < np; ++k) { uint32_t a = rd1<uint32_t>(in), b = rd1<uint32_t>(in); nd.points3d[a] = b; }
        const int nn = rd1<int32_t>(in);
        nd.neighbours = rd<uint32_t>(in, nn);
        graph.nodes[nd.id] = nd;
    }
    const int nPts = rd1<int32_t>(in);
    for (int j = 0; j < nPts; ++j) {
        const uint32_t id = rd1<uint32_t>(in);
        MapPointView mp;
        auto p = rd<double>(in, 3);
        std::copy(p.begin(), p.end(), mp.point3d);
        mp.status = rd1<int32_t>(in) != 0; mp.observers = rd1<uint32_t>(in);
        map.points[id] = mp;
    }
    auto K = rd<double>(in, 9);
    const int current = rd1<int32_t>(in);
#ifdef EACHAM_TEST_GLUE
    // the same fixture as reference-shaped objects; RefineBA(currentFrameId, graph, map, K, config) as the app calls it
    (void)kRefineBA;
    auto rgraph = std::make_shared<eacham::graph_t>();
    auto rmap = std::make_shared<eacham::Map>();
    for (auto& kv : graph.nodes) {
        auto* node = rgraph->TestCreate(kv.first);
        node->TestSetValid(kv.second.valid);
        if (kv.second.fixed) rgraph->FixNode(kv.first);
        Eigen::Matrix4d M;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) M(r, c) = kv.second.transform[4 * r + c];
        node->SetTransform(M);
        std::vector<cv::Point2f> kps(kv.second.keypoints.size() / 2);
        for (size_t k = 0; k < kps.size(); ++k) kps[k].x = kv.second.keypoints[2 * k], kps[k].y = kv.second.keypoints[2 * k + 1];
        node->TestSetFeatures(kps);
        for (auto& p : kv.second.points3d) node->SetPoint3d(p.first, p.second, false);
        for (unsigned nb : kv.second.neighbours) node->TestFactor(nb);
    }
    for (auto& kv : map.points) {
        auto& mp = rmap->TestInsert(kv.first);
        mp.point3d = Eigen::Vector3d(kv.second.point3d[0], kv.second.point3d[1], kv.second.point3d[2]);
        mp.isValid = kv.second.status;
        for (unsigned k = 0; k < kv.second.observers; ++k) mp.observers[1000 + k] = 0;   // only the COUNT is read (:109)
    }
    cv::Mat Kmat;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Kmat.at<double>(r, c) = K[3 * r + c];
    eacham::OptimizerConfig rcfg{"LM", 100, 1e-5f, 10.0f, false};
    RefineBAReport rep = glue::RefineBA(current, rgraph, rmap, Kmat, rcfg);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) K[3 * r + c] = Kmat.at<double>(r, c);
    for (auto& kv : graph.nodes)      // read the objects back into the views the common tail writes out
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) kv.second.transform[4 * r + c] = rgraph->Get(kv.first)->GetTransform()(r, c);
    for (auto& kv : map.points) {
        const Eigen::Vector3d X = rmap->Get(kv.first);
        kv.second.point3d[0] = X(0), kv.second.point3d[1] = X(1), kv.second.point3d[2] = X(2);
        kv.second.status = rmap->GetStatus(kv.first);
    }
#else
    OptimizerConfig cfg; cfg.method = "LM"; cfg.maxIter = 100; cfg.maxTolerance = 1e-5f;
    RefineBAReport rep = RefineBA(matcher.context().get(), current, graph, map, K.data(), cfg);
#endif
    std::vector<double> meta = {(double)rep.skipped, rep.initial_error, rep.final_error, (double)rep.outer_iterations,
                                (double)rep.inner_iterations, (double)rep.frames, (double)rep.landmarks, (double)rep.observations};
    wr(out, meta); wr(out, K);
    std::vector<double> Ts, Ps;
    for (auto& kv : graph.nodes) Ts.insert(Ts.end(), kv.second.transform, kv.second.transform + 16);
    std::vector<uint32_t> ids;
    for (auto& kv : map.points) ids.push_back(kv.first);
    std::sort(ids.begin(), ids.end());
    std::vector<int32_t> status;
    for (uint32_t id : ids) { Ps.insert(Ps.end(), map.points[id].point3d, map.points[id].point3d + 3); status.push_back(map.points[id].status); }
    wr(out, Ts); wr(out, Ps); wr(out, status);
    std::printf("adapter driver ok: %zu frames, %zu landmarks, %zu observations\n", rep.frames, rep.landmarks, rep.observations);
    return 0;
}
