// The incremental loop of apps/sfm/main.cpp:76-240 on the device library, written against the reference-typed entry points:
//   MatchAllPairs + Graph::Connect both ways (:79-147), FindBestPair (:161-162), then
//   while (GetBestPairForValid) { RecoverPosePnP -> TriangulateFrame(2) -> RefineBA -> TriangulateFrame(3) } (:188-214),
//   global RefineBA(-1) (:216-220).
// Graph / Node / Map / cv::Mat are the test stand-ins of ref_standins.hpp (OpenCV, Eigen and the reference's headers are not
// in the image); the functions called are the drop-in definitions of ReferenceGlue.hpp / ReconstructionHip.hpp.
//   sfm_loop_driver <in.bin> <out.bin>
// in : F, dim; per frame n, keypoints (float n x 2), descriptors (float n x dim); K (9 doubles); seven config floats
//      [+ an eighth: pixel threshold for the H / E inlier counts of RecoverPoseTwoView, 0 = the reference's LMedS masks]
// out: per frame valid flag + 16 doubles (world -> camera); map: n points x (id, x, y, z, valid, observers); log counters
#include <chrono>
#include <cstring>
#include <cstdio>
#include <fstream>
#include <set>

#include "ref_standins.hpp"
#include "eacham/FeatureMatcherHip.hpp"
#include "eacham/ReconstructionHip.hpp"
#include "eacham/ReferenceGlue.hpp"

using namespace eacham;
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
    const int F = rd1<int32_t>(in), dim = rd1<int32_t>(in);
    auto graph = std::make_shared<graph_t>();
    auto globalMap = std::make_shared<Map>();
    std::vector<std::vector<float>> store;
    std::vector<DescriptorView> frames;
    for (int f = 0; f < F; ++f) {
        const int n = rd1<int32_t>(in);
        const auto kp = rd<float>(in, 2 * (size_t)n);
        store.push_back(rd<float>(in, (size_t)n * dim));
        frames.push_back(DescriptorView{store.back().data(), n, dim});
        std::vector<cv::Point2f> pts(n);
        for (int k = 0; k < n; ++k) pts[k].x = kp[2 * k], pts[k].y = kp[2 * k + 1];
        graph->TestCreate((unsigned)f)->TestSetFeatures(pts);   // graph->Create(frame.id, features, descriptors, image), :71-75
    }
    const auto K9 = rd<double>(in, 9);
    const float inliersRatio = rd1<float>(in), initialMaxReprError = rd1<float>(in), initialMinTriAngle = rd1<float>(in);
    const float maxReprError = rd1<float>(in), minTriAngle = rd1<float>(in);
    const int minPnpInliers = (int)rd1<float>(in);
    const unsigned initialMinInliers = (unsigned)rd1<float>(in);
    float inlierPx = 0.0f;
    in.read((char*)&inlierPx, sizeof(float));
    if (!in) inlierPx = 0.0f;
    cv::Mat K;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) K.at<double>(r, c) = K9[3 * r + c];
    const eacham::OptimizerConfig refineOpt{"LM", 100, 1e-5f, 10.0f, false}, globalOpt{"LM", 50, 1e-4f, 10.0f, false};

    Context& ctx = glue::shared_context();   // one device context for matching, reconstruction, triangulation and BA
    // ---- match features (:79-147): every unordered pair once, both directions + mutual check + the 30 / 30 thresholds on the device
    std::vector<std::pair<unsigned, unsigned>> pairs;
    for (unsigned i = 0; i < (unsigned)F; ++i)
        for (unsigned j = i + 1; j < (unsigned)F; ++j) pairs.push_back({i, j});
    const auto t0 = std::chrono::steady_clock::now();
    const MatchGraph g = MatchAllPairs(ctx, frames, pairs, inliersRatio, 30, 30);
    const auto t1 = std::chrono::steady_clock::now();
    size_t edges = 0;
    for (size_t p = 0; p < pairs.size(); ++p) {
        if (g.counts[p] <= 0) continue;
        match_t m12, m21;
        for (int64_t k = g.offsets[p]; k < g.offsets[p] + g.counts[p]; ++k) m12[g.q[k]] = g.t[k], m21[g.t[k]] = g.q[k];
        graph->Connect(graph->Get(pairs[p].first), graph->Get(pairs[p].second), std::move(m12));   // :144-145
        graph->Connect(graph->Get(pairs[p].second), graph->Get(pairs[p].first), std::move(m21));
        ++edges;
    }
    // ---- two view (:157-176)
    const Sampling sampling = (argc > 3 && !std::strcmp(argv[3], "counter")) ? Sampling::Counter : Sampling::OpenCV;
    glue::ReconstructionManagerHip<graph_t, Map> reconstructor(ctx, graph, globalMap, initialMaxReprError, initialMinTriAngle, minPnpInliers, 12345, inlierPx,
                                                                sampling);
    auto [prevId, currentId] = glue::FindBestPair(graph, globalMap, reconstructor, K, initialMinInliers);
    std::vector<double> log{(double)edges, (double)prevId, (double)currentId, (double)globalMap->GetAll().size()};
    if (prevId > graph->Size() || currentId > graph->Size()) {
        wr(out, log);
        std::printf("sfm loop: no initial pair\n");
        return 3;
    }
    // ---- the incremental loop (:178-214)
    // The view-graph query on the RESIDENT match graph: uploaded once, then only the frames the loop has just changed are refreshed —
    // the one it posed and triangulated and that frame's factor neighbours (TriangulateFrame's SetPoint3d reaches no further).
    std::vector<size_t> kpCount(F);
    for (int f = 0; f < F; ++f) kpCount[f] = graph->Get((unsigned)f)->GetFeatures().size();
    ResidentMatchGraph rg(ctx, pairs, g, kpCount);
    auto state_of = [&](unsigned f) {
        auto* n = graph->Get(f);
        ResidentMatchGraph::FrameState st{f, n->IsValid(), std::vector<uint8_t>(n->GetFeatures().size(), 0)};
        for (const auto& kv : n->GetPoints3d()) st.has3d[kv.first] = !n->IsPoint3dTwoView(kv.first);
        return st;
    };
    {
        std::vector<ResidentMatchGraph::FrameState> all;
        for (int f = 0; f < F; ++f) all.push_back(state_of((unsigned)f));
        rg.SetFrames(all);
    }
    auto best_pair = [&](const std::set<unsigned>& excluded) { return rg.Query(excluded); };
    std::set<unsigned> excluded{prevId, currentId};
    BestPair bp = best_pair(excluded);
    int pnp_ok = 0, pnp_failed = 0;
    double ms_pnp = 0, ms_tri = 0, ms_ba = 0, ms_query = 0;
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    while (bp.id <= graph->Size() && bp.id2 <= graph->Size()) {
        auto ta = std::chrono::steady_clock::now();
        const bool posed = reconstructor.RecoverPosePnP(bp.id, bp.id2, K);
        ms_pnp += since(ta);
        if (posed) {
            ta = std::chrono::steady_clock::now();
            eacham::TriangulateFrame(bp.id2, graph, globalMap, K, 2, maxReprError, minTriAngle);   // :203
            ms_tri += since(ta);
            ta = std::chrono::steady_clock::now();
            eacham::RefineBA((int)bp.id2, graph, globalMap, K, refineOpt);                           // :207
            ms_ba += since(ta);
            ta = std::chrono::steady_clock::now();
            eacham::TriangulateFrame(bp.id2, graph, globalMap, K, 3, maxReprError, minTriAngle);   // :209
            ms_tri += since(ta);
            excluded = {};
            ++pnp_ok;
        } else {
            ++pnp_failed;
        }
        ta = std::chrono::steady_clock::now();
        {
            std::vector<ResidentMatchGraph::FrameState> changed{state_of(bp.id2)};
            for (const auto& f : graph->Get(bp.id2)->GetFactors()) changed.push_back(state_of(f.first));
            rg.SetFrames(changed);
        }
        bp = best_pair(excluded);
        ms_query += since(ta);
        if (bp.id > graph->Size() || bp.id2 > graph->Size()) break;
        excluded.insert(bp.id);
        excluded.insert(bp.id2);
    }
    const auto tg = std::chrono::steady_clock::now();
    eacham::RefineBA(-1, graph, globalMap, K, globalOpt);                                            // :216-220
    const double ms_global = since(tg);
    const auto t2 = std::chrono::steady_clock::now();
    const double match_ms = std::chrono::duration<double, std::milli>(t1 - t0).count(), sfm_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    // ---- results
    log.push_back(pnp_ok), log.push_back(pnp_failed);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) log.push_back(K.at<double>(r, c));
    wr(out, log);
    std::vector<double> poses;
    for (int f = 0; f < F; ++f) {
        auto* n = graph->Get((unsigned)f);
        poses.push_back(n->IsValid());
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) poses.push_back(n->GetTransform()(r, c));
    }
    wr(out, poses);
    std::vector<double> pts;
    for (const auto& kv : globalMap->GetAll()) {
        pts.push_back(kv.first);
        for (int e = 0; e < 3; ++e) pts.push_back(kv.second.point3d(e));
        pts.push_back(kv.second.isValid);
        pts.push_back((double)kv.second.observers.size());
    }
    wr(out, pts);
    if (inlierPx > 0.0f) std::printf("H/E rule: inliers at %.1f px (not the reference's); sampling: %s\n", inlierPx, sampling == Sampling::OpenCV ? "opencv" : "counter");
    else std::printf("H/E rule: LMedS masks (the reference's); sampling: %s\n", sampling == Sampling::OpenCV ? "opencv" : "counter");
    std::printf("sfm loop ok: %zu edges, initial pair %u-%u, %d frames added, %d PnP failures, %zu map points; [Match] %.1f ms (upload + %zu pairs), [SfM] %.1f ms = PnP %.1f + TriangulateFrame %.1f + RefineBA %.1f + GetBestPairForValid %.1f + global BA %.1f + two-view\n",
                edges, prevId, currentId, pnp_ok, pnp_failed, globalMap->GetAll().size(), match_ms, pairs.size(), sfm_ms, ms_pnp, ms_tri, ms_ba, ms_query, ms_global);
    return 0;
}
