// ref_standins.hpp — TEST-ONLY stand-ins for the types the reference-typed glue is written against
// (include/eacham/ReferenceGlue.hpp). OpenCV, Eigen and eacham's own headers are not in this image, so the glue cannot
// be compiled against them here; what it needs from them is a handful of accessor NAMES:
//   cv::Mat::at<double>(r, c), cv::Point2f{x, y}
//   Eigen::Matrix4d / Vector3d element access  M(r, c), v(i)
//   Graph::GetNodes / Get / IsFixed, Node::IsValid / GetTransform / SetTransform / GetFeatures / GetPoints3d /
//   SetPoint3d / GetFactors (Factor::matches), Map::Get / GetStatus / GetObservers / GetAll / Add / UpdatePoint /
//   UpdateStatus / AddObserver / RemoveObserver, OptimizerConfig's five fields
//   and for ReconstructionHip.hpp: Node::GetFactor / GetKeyPoint / HasPoint3d / IsPoint3dTwoView / GetPoint3d / SetValid,
//   Factor::quality / transform, Graph::Connect / Size, Map::Add(point, colour)
//       modules/sfm/data/Graph.h:44-57,108-116, Node.h:24-31,57-71,100-103,126-134,146-149,204-207,
//       Map.h:15-23,40-49,59-71,87-99,101-127,129-177,179-196, modules/sfm/config/SfmConfig.h:15-22 (under /root/reference)
// These classes carry those names with the smallest bodies that make them work, plus a few Test* methods the drivers use
// to load a fixture. They are not a port of the reference's containers (no mutexes, no images, no quality bookkeeping).
#pragma once

#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#define EACHAM_GLUE_STANDINS 1

namespace cv {
struct Point2f { float x = 0, y = 0; };
class Mat {  // a 3x3 matrix of doubles is all the glue touches
public:
    Mat() : v_(9, 0.0) {}
    template <class T> T& at(int r, int c) { return v_[3 * r + c]; }
    template <class T> const T& at(int r, int c) const { return v_[3 * r + c]; }
private:
    std::vector<double> v_;
};
}  // namespace cv

namespace Eigen {
class Vector3d {
public:
    Vector3d() = default;
    Vector3d(double x, double y, double z) : v_{x, y, z} {}
    double& operator()(int i) { return v_[i]; }
    double operator()(int i) const { return v_[i]; }
private:
    double v_[3] = {0, 0, 0};
};
class Matrix4d {  // column-major like Eigen's default: the glue must not rely on the storage order
public:
    double& operator()(int r, int c) { return v_[4 * c + r]; }
    double operator()(int r, int c) const { return v_[4 * c + r]; }
private:
    double v_[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};
}  // namespace Eigen

namespace eacham {

using match_t = std::unordered_map<unsigned, unsigned>;

struct Factor {
    unsigned id = 0;
    float quality = 0;
    match_t matches;
    unsigned points3dCount = 0;
    Eigen::Matrix4d transform;
};

struct OptimizerConfig {
    std::string method;
    int maxIter;
    float maxTolerance;
    float delta;
    bool usePreconditioner;
};

template <class FT, class DT>
class Node {
public:
    explicit Node(unsigned id) : id_(id) {}
    void SetTransform(const Eigen::Matrix4d& t) { transform_ = t; }
    const Eigen::Matrix4d& GetTransform() const { return transform_; }
    const FT& GetFeatures() { return keypoints_; }
    const std::unordered_map<unsigned, unsigned>& GetPoints3d() { return points3d_; }
    void SetPoint3d(unsigned id2d, unsigned id3d, bool isTwoView) { points3d_[id2d] = id3d; twoView_[id2d] = isTwoView; }
    const std::unordered_map<unsigned, Factor>& GetFactors() { return factors_; }
    bool IsValid() const { return valid_; }
    void SetValid(bool v) { valid_ = v; }
    unsigned GetId() const { return id_; }
    Factor& GetFactor(unsigned id) {
        auto it = factors_.find(id);
        if (it == factors_.end()) throw std::runtime_error("Node::GetFactor: factor is not found");
        return it->second;
    }
    const cv::Point2f& GetKeyPoint(unsigned id) const { return keypoints_.at(id); }
    bool HasPoint3d(unsigned id2d) const { return points3d_.count(id2d) > 0; }
    bool IsPoint3dTwoView(unsigned id2d) const { return twoView_.at(id2d); }
    unsigned GetPoint3d(unsigned id2d) const { return points3d_.at(id2d); }
    // test-only loaders
    void TestSetValid(bool v) { valid_ = v; }
    void TestSetFeatures(FT k) { keypoints_ = std::move(k); }
    Factor& TestFactor(unsigned other) { factors_[other].id = other; return factors_[other]; }
private:
    unsigned id_;
    bool valid_ = false;
    FT keypoints_;
    std::unordered_map<unsigned, Factor> factors_;
    std::unordered_map<unsigned, unsigned> points3d_;
    std::unordered_map<unsigned, bool> twoView_;
    Eigen::Matrix4d transform_;
};

template <class FT, class DT>
class Graph {
public:
    ~Graph() { for (auto& kv : nodes_) delete kv.second; }
    Node<FT, DT>* Get(unsigned id) {
        auto it = nodes_.find(id);
        return it == nodes_.end() ? nullptr : it->second;
    }
    const std::map<unsigned, Node<FT, DT>*>& GetNodes() { return nodes_; }
    void Connect(Node<FT, DT>* node1, Node<FT, DT>* node2, match_t&& matches) {
        Factor& f = node1->TestFactor(node2->GetId());
        f.quality = (float)matches.size();
        f.matches = std::move(matches);
    }
    size_t Size() const { return nodes_.size(); }
    void FixNode(unsigned id) { fixed_.insert(id); }
    bool IsFixed(unsigned id) { return fixed_.count(id) > 0; }
    Node<FT, DT>* TestCreate(unsigned id) {
        if (!nodes_.count(id)) nodes_[id] = new Node<FT, DT>(id);
        return nodes_[id];
    }
private:
    std::map<unsigned, Node<FT, DT>*> nodes_;
    std::set<unsigned> fixed_;
};

using descriptor_tt = cv::Mat;
using graph_t = Graph<std::vector<cv::Point2f>, descriptor_tt>;
using node_t = Node<std::vector<cv::Point2f>, descriptor_tt>;

struct MapPointData {
    unsigned id = 0;
    Eigen::Vector3d point3d;
    bool isValid = false;
    std::unordered_map<unsigned, unsigned> observers;
};

class Map {
public:
    unsigned Add(const Eigen::Vector3d& p) {
        ++counter_;
        MapPointData d;
        d.id = counter_;
        d.point3d = p;
        points_[counter_] = d;
        return counter_;
    }
    unsigned Add(const Eigen::Vector3d& p, const Eigen::Vector3d& /*colour*/) { return Add(p); }
    void UpdatePoint(unsigned id, const Eigen::Vector3d& p) { at(id).point3d = p; }
    void UpdateStatus(unsigned id, bool valid) { at(id).isValid = valid; }
    Eigen::Vector3d Get(unsigned id) const { return at(id).point3d; }
    bool GetStatus(unsigned id) const { return at(id).isValid; }
    void AddObserver(unsigned frame, unsigned point2d, unsigned point3d) { at(point3d).observers[frame] = point2d; }
    void RemoveObserver(unsigned frame, unsigned /*point2d*/, unsigned point3d) { at(point3d).observers.erase(frame); }
    std::unordered_map<unsigned, unsigned> GetObservers(unsigned id) const { return at(id).observers; }
    const std::unordered_map<unsigned, MapPointData>& GetAll() const { return points_; }
    // test-only loader: a point under a given id (the reference only hands out consecutive ids)
    MapPointData& TestInsert(unsigned id) {
        points_[id].id = id;
        if (id > counter_) counter_ = id;
        return points_[id];
    }
private:
    MapPointData& at(unsigned id) {
        auto it = points_.find(id);
        if (it == points_.end()) throw std::runtime_error("Map: point is not found");
        return it->second;
    }
    const MapPointData& at(unsigned id) const {
        auto it = points_.find(id);
        if (it == points_.end()) throw std::runtime_error("Map: point is not found");
        return it->second;
    }
    std::unordered_map<unsigned, MapPointData> points_;
    unsigned counter_ = 0;
};

}  // namespace eacham
