// Test driver for include/eacham/TriangulatorHip.hpp: reads a binary fixture written by
// tests/test_cpp_adapters.py, runs TriangulatePointRansac and TriangulateFrame through the C-ABI and
// writes graph/map state back for comparison with a walk done in Python on the oracle.
#include <cstdio>
#include <fstream>
#include <vector>

#ifdef EACHAM_TEST_GLUE  // TriangulateFrame goes through the reference-typed glue on stand-ins of Graph / Node / Map / cv::Mat
#include "ref_standins.hpp"
#include "eacham/ReferenceGlue.hpp"
static void (*const kTriangulateFrame)(const unsigned, std::shared_ptr<eacham::graph_t>, std::shared_ptr<eacham::Map>, const cv::Mat&,
                                       const unsigned, const float, const float) = &eacham::TriangulateFrame;  // Triangulator.h:41-43
#endif
#include "eacham/TriangulatorHip.hpp"

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
    Context ctx(0);
    TriGraphView graph;
    TriMapView map;
    const int nNodes = rd1<int32_t>(in);
    for (int i = 0; i < nNodes; ++i) {
        TriNodeView nd;
        const uint32_t id = rd1<uint32_t>(in);
        nd.valid = rd1<int32_t>(in) != 0;
        auto T = rd<double>(in, 16);
        std::copy(T.begin(), T.end(), nd.transform);
        const int nk = rd1<int32_t>(in);
        nd.keypoints = rd<float>(in, 2 * (size_t)nk);
        const int np = rd1<int32_t>(in);
        for (int k = 0; k < np; ++k) { uint32_t a = rd1<uint32_t>(in), b = rd1<uint32_t>(in); nd.points3d[a] = b; }
        const int nf = rd1<int32_t>(in);
        for (int f = 0; f < nf; ++f) {
            const uint32_t other = rd1<uint32_t>(in);
            const int nm = rd1<int32_t>(in);
            auto flat = rd<uint32_t>(in, 2 * (size_t)nm);
            auto& v = nd.factors[other];
            for (int k = 0; k < nm; ++k) v.push_back({flat[2 * k], flat[2 * k + 1]});
        }
        graph.nodes[id] = nd;
    }
    const int nPts = rd1<int32_t>(in);
    for (int j = 0; j < nPts; ++j) {
        const uint32_t id = rd1<uint32_t>(in);
        TriMapPoint mp;
        auto p = rd<double>(in, 3);
        std::copy(p.begin(), p.end(), mp.point3d);
        mp.isValid = rd1<int32_t>(in) != 0;
        const int no = rd1<int32_t>(in);
        for (int k = 0; k < no; ++k) { uint32_t a = rd1<uint32_t>(in), b = rd1<uint32_t>(in); mp.observers[a] = b; }
        map.points[id] = mp;
        if (id > map.mapPointId) map.mapPointId = id;
    }
    auto K = rd<double>(in, 9);
    const uint32_t frameId = rd1<uint32_t>(in), minObservers = rd1<uint32_t>(in);
    const float maxErr = rd1<float>(in), minAngle = rd1<float>(in);

    // one TriangulatePointRansac call on the first factor's first three matches-worth of data
    const int nd = rd1<int32_t>(in);
    std::vector<EstimatorData> data(nd);
    for (auto& d : data) {
        auto T = rd<double>(in, 16);
        std::copy(T.begin(), T.end(), d.transform);
        auto p = rd<double>(in, 2);
        d.point2d[0] = p[0]; d.point2d[1] = p[1];
    }
    std::vector<double> X(3);
    std::vector<bool> inl;
    const bool ok = TriangulatePointRansac(ctx, data, K.data(), X.data(), inl, maxErr, minAngle);
    std::vector<int32_t> single{ok ? 1 : 0};
    for (bool b : inl) single.push_back(b ? 1 : 0);
    wr(out, single); wr(out, X);

#ifdef EACHAM_TEST_GLUE
    (void)kTriangulateFrame;
    auto rgraph = std::make_shared<eacham::graph_t>();
    auto rmap = std::make_shared<eacham::Map>();
    for (auto& kv : graph.nodes) {
        auto* node = rgraph->TestCreate(kv.first);
        node->TestSetValid(kv.second.valid);
        Eigen::Matrix4d M;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) M(r, c) = kv.second.transform[4 * r + c];
        node->SetTransform(M);
        std::vector<cv::Point2f> kps(kv.second.keypoints.size() / 2);
        for (size_t k = 0; k < kps.size(); ++k) kps[k].x = kv.second.keypoints[2 * k], kps[k].y = kv.second.keypoints[2 * k + 1];
        node->TestSetFeatures(kps);
        for (auto& p : kv.second.points3d) node->SetPoint3d(p.first, p.second, false);
        for (auto& f : kv.second.factors) {
            auto& fac = node->TestFactor(f.first);
            for (auto& mm : f.second) fac.matches[mm.first] = mm.second;
        }
    }
    for (auto& kv : map.points) {
        auto& mp = rmap->TestInsert(kv.first);
        mp.point3d = Eigen::Vector3d(kv.second.point3d[0], kv.second.point3d[1], kv.second.point3d[2]);
        mp.isValid = kv.second.isValid;
        for (auto& o : kv.second.observers) mp.observers[o.first] = o.second;
    }
    cv::Mat Kmat;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Kmat.at<double>(r, c) = K[3 * r + c];
    TriangulateFrameReport rep = glue::TriangulateFrame(frameId, rgraph, rmap, Kmat, minObservers, maxErr, minAngle);
    // the objects back into the views the common tail writes out
    for (auto& kv : graph.nodes) {
        kv.second.points3d.clear();
        for (auto& p : rgraph->Get(kv.first)->GetPoints3d()) kv.second.points3d[p.first] = p.second;
    }
    map.points.clear();
    map.mapPointId = 0;
    for (auto& kv : rmap->GetAll()) {
        TriMapPoint mp;
        mp.point3d[0] = kv.second.point3d(0), mp.point3d[1] = kv.second.point3d(1), mp.point3d[2] = kv.second.point3d(2);
        mp.isValid = kv.second.isValid;
        for (auto& o : kv.second.observers) mp.observers[o.first] = o.second;
        map.points[kv.first] = mp;
        if (kv.first > map.mapPointId) map.mapPointId = kv.first;
    }
#else
    TriangulateFrameReport rep = TriangulateFrame(ctx, frameId, graph, map, K.data(), minObservers, maxErr, minAngle);
#endif
    std::vector<uint32_t> meta{rep.total, rep.added, rep.reobserved, map.mapPointId};
    wr(out, meta);
    for (auto& kv : graph.nodes) {   // per node (ascending id): points3d as flat pairs
        std::vector<uint32_t> flat;
        for (auto& p : kv.second.points3d) { flat.push_back(p.first); flat.push_back(p.second); }
        wr(out, flat);
    }
    std::vector<uint32_t> ids, valid, obs;
    std::vector<double> P;
    for (auto& kv : map.points) {
        ids.push_back(kv.first);
        valid.push_back(kv.second.isValid);
        P.insert(P.end(), kv.second.point3d, kv.second.point3d + 3);
        obs.push_back((uint32_t)kv.second.observers.size());
        for (auto& o : kv.second.observers) { obs.push_back(o.first); obs.push_back(o.second); }
    }
    wr(out, ids); wr(out, valid); wr(out, P); wr(out, obs);
    std::printf("tri driver ok: total %u added %u reobserved %u\n", rep.total, rep.added, rep.reobserved);
    return 0;
}
