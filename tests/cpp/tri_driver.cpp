// Test driver for include/eacham/TriangulatorHip.hpp: reads a binary fixture written by
// tests/test_cpp_adapters.py, runs TriangulatePointRansac and TriangulateFrame through the C-ABI and
// writes graph/map state back for comparison with a walk done in Python on the oracle.
#include <cstdio>
#include <fstream>
#include <vector>

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

    TriangulateFrameReport rep = TriangulateFrame(ctx, frameId, graph, map, K.data(), minObservers, maxErr, minAngle);
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
