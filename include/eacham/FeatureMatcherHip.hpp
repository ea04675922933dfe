// FeatureMatcherHip.hpp — C++ adapter that restates eacham's matcher interface on top of the C-ABI.
//
//   IFeatureMatcher<T>::Match(const T&, const T&) -> std::unordered_map<unsigned, unsigned>
//       /root/reference/modules/base/features/IFeatureMatcher.h:8-20
//   FeatureMatcherFlann(const float inliersRatio); Match(const cv::Mat&, const cv::Mat&)
//       /root/reference/modules/base/features/FeatureMatcherFlann.h:11-25, .cpp:8-30
//
// Header-only; link against libeacham_hip.so. Errors become std::runtime_error like the reference's
// own accessors (Node.h:78-98); the C-ABI itself never throws. Thread-safe: one shared instance may
// be called from many threads (apps/sfm/main.cpp:98-109) — concurrent calls are combined into batches.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <exception>
#include <functional>
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
    // Bumped by whoever rewrites the context's descriptor store wholesale (MatchAllPairs): a FeatureMatcherHip
    // that caches uploads on the same context then knows its slots are gone.
    unsigned store_generation() const { return generation_.load(); }
    void store_rewritten() { generation_.fetch_add(1); }

private:
    eacham_ctx* ctx_ = nullptr;
    std::atomic<unsigned> generation_{0};
};

// Drop-in for eacham::FeatureMatcherFlann. The reference calls Match(d1, d2) once per ORDERED frame pair, from many
// threads at once on one shared instance, always with the descriptor matrices the Nodes own
// (apps/sfm/main.cpp:98-109: `std::async(&FeatureMatcherFlann::Match, &matcher, node1->GetDescriptors(), ...)`).
// Two things keep that call pattern off the PCIe bus and the launch latency:
//   * uploads are cached by (data pointer, rows, dim) + a sampled fingerprint of the values: F frames are uploaded
//     once and serve all F (F - 1) calls (LRU over `cacheCapacity` device-resident frames);
//   * concurrent callers are combined: one of the waiting threads takes every request queued so far and runs them as
//     ONE eacham_match_pairs_directed launch sequence, the others sleep until their result is ready.
// Integer-valued descriptors (SIFT) take the exact int8 path; the first frame that is not integer-valued switches
// the instance to the fp32 MFMA path for good (one extractor feeds one matcher in the reference).
class FeatureMatcherHip : public IFeatureMatcher<DescriptorView> {
public:
    // `inliersRatio` is kept for signature parity; as in the reference it is NOT the ratio of the
    // test, which is the literal 0.8 (FeatureMatcherFlann.cpp:23) unless `ratio` overrides it.
    explicit FeatureMatcherHip(float inliersRatio = 0.8f, int device = 0, double ratio = 0.8, int cacheCapacity = 4096)
        : inliersRatio_(inliersRatio), ratio_(ratio), ctx_(device),
          capacity_(cacheCapacity < 2 * kMaxBatch ? 2 * kMaxBatch : cacheCapacity) {}

    MatchType Match(const DescriptorView& d1, const DescriptorView& d2) override {
        Request req;
        req.d1 = d1;
        req.d2 = d2;
        std::unique_lock<std::mutex> lk(mu_);
        queue_.push_back(&req);
        while (!req.done) {
            if (leader_active_) {
                cv_.wait(lk);
                continue;
            }
            // this thread serves everything queued so far (its own request included unless an earlier leader took it)
            leader_active_ = true;
            std::vector<Request*> batch;
            const size_t take = queue_.size() < (size_t)kMaxBatch ? queue_.size() : (size_t)kMaxBatch;
            batch.assign(queue_.begin(), queue_.begin() + take);
            queue_.erase(queue_.begin(), queue_.begin() + take);
            lk.unlock();
            try {
                run_batch(batch);
            } catch (...) {  // a failure of the batch as a whole: every member sees it
                for (Request* r : batch)
                    if (!r->error) r->error = std::current_exception();
            }
            lk.lock();
            for (Request* r : batch) r->done = true;
            leader_active_ = false;
            cv_.notify_all();
        }
        lk.unlock();
        if (req.error) std::rethrow_exception(req.error);
        return std::move(req.result);
    }

#ifdef EACHAM_HIP_HAVE_OPENCV
    // Drop-in for FeatureMatcherFlann::Match(const cv::Mat&, const cv::Mat&): CV_32F, continuous.
    MatchType Match(const cv::Mat& d1, const cv::Mat& d2) {
        if (d1.type() != CV_32F || d2.type() != CV_32F || !d1.isContinuous() || !d2.isContinuous())
            throw std::runtime_error("eacham_hip: descriptors must be continuous CV_32F matrices");
        return Match(DescriptorView{d1.ptr<float>(), d1.rows, d1.cols}, DescriptorView{d2.ptr<float>(), d2.rows, d2.cols});
    }
#endif

    // Forgets every cached upload.
    void ClearCache() {
        std::unique_lock<std::mutex> lk(mu_);
        while (leader_active_) cv_.wait(lk);
        drop_cache(true);
    }
    // A cached upload is recognised by (data pointer, rows, dim) + a fingerprint of the values. By default the
    // fingerprint SAMPLES the matrix (1024 words + the last): a buffer that is freed and reused with other values
    // of the same shape that happen to agree on every sample would be matched as the OLD frame — silently, on a path
    // whose claim is bit-exactness. Two ways to close that hole:
    //   * SetFullContentCheck(true): the fingerprint covers EVERY word of the matrix on every call (about 0.1 ms per
    //     2000 x 128 frame per call on the host): no reuse can go unnoticed, at the price of reading the descriptors
    //     once per Match();
    //   * Invalidate(data): the caller states that the buffer at `data` was rewritten or freed (what eacham's
    //     Node::SetDescriptors amounts to): the next Match() with that address uploads it again.
    void SetFullContentCheck(bool on) {
        std::unique_lock<std::mutex> lk(mu_);
        while (leader_active_) cv_.wait(lk);
        if (on != full_check_) drop_cache(false);  // fingerprints of the two kinds do not compare
        full_check_ = on;
    }
    void Invalidate(const float* data) {
        std::unique_lock<std::mutex> lk(mu_);
        while (leader_active_) cv_.wait(lk);
        for (auto it = cache_.begin(); it != cache_.end();) {
            if (it->first.data == data) {
                free_slots_.push_back(it->second.slot);
                it = cache_.erase(it);
            } else {
                ++it;
            }
        }
    }
    struct Stats {
        uint64_t calls = 0, batches = 0, uploads = 0, cache_hits = 0;
    };
    Stats stats() {
        std::lock_guard<std::mutex> lk(mu_);
        return stats_;
    }

    Context& context() { return ctx_; }

private:
    static constexpr int kMaxBatch = 256;  // requests combined into one launch sequence
    struct Request {
        DescriptorView d1, d2;
        MatchType result;
        std::exception_ptr error;
        bool done = false;
    };
    struct Key {
        const float* data;
        int rows, dim;
        bool operator==(const Key& o) const { return data == o.data && rows == o.rows && dim == o.dim; }
    };
    struct KeyHash {
        size_t operator()(const Key& k) const {
            return std::hash<const void*>()(k.data) ^ (std::hash<int>()(k.rows) * 0x9E3779B97F4A7C15ull) ^ ((size_t)k.dim << 48);
        }
    };
    struct Entry {
        int slot;
        uint64_t fingerprint, last_use;
    };

    // FNV-1a over <= 1024 words spread evenly over the matrix, plus its last word
    static uint64_t fingerprint(const DescriptorView& v) {
        const size_t total = (size_t)(v.rows > 0 ? v.rows : 0) * (size_t)(v.dim > 0 ? v.dim : 0);
        uint64_t h = 1469598103934665603ull ^ total;
        if (!total) return h;
        const size_t step = total / 1024 ? total / 1024 : 1;
        auto mix = [&](size_t i) {
            uint32_t w;
            std::memcpy(&w, v.data + i, sizeof(w));
            h = (h ^ w) * 1099511628211ull;
        };
        for (size_t i = 0; i < total; i += step) mix(i);
        mix(total - 1);
        return h;
    }

    // every word of the matrix: four interleaved multiply-xorshift lanes over 8-byte words (runs at memory speed)
    static uint64_t full_fingerprint(const DescriptorView& v) {
        const size_t total = (size_t)(v.rows > 0 ? v.rows : 0) * (size_t)(v.dim > 0 ? v.dim : 0);
        uint64_t h[4] = {0x9E3779B97F4A7C15ull ^ total, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
        const size_t words = total / 2;  // 8-byte words
        const char* p = reinterpret_cast<const char*>(v.data);
        for (size_t i = 0; i < words; ++i) {
            uint64_t w;
            std::memcpy(&w, p + 8 * i, 8);
            uint64_t& x = h[i & 3];
            x = (x ^ w) * 0xFF51AFD7ED558CCDull;
            x ^= x >> 29;
        }
        if (total & 1) {
            uint32_t w;
            std::memcpy(&w, v.data + total - 1, 4);
            h[0] = (h[0] ^ w) * 0xC4CEB9FE1A85EC53ull;
        }
        uint64_t r = h[0];
        for (int k = 1; k < 4; ++k) r = (r ^ (h[k] + 0x9E3779B97F4A7C15ull + (r << 6) + (r >> 2))) * 0xFF51AFD7ED558CCDull;
        return r ^ (r >> 32);
    }

    void drop_cache(bool clear_device) {
        cache_.clear();
        free_slots_.clear();
        next_slot_ = 0;
        if (clear_device) ctx_.check(eacham_clear_descriptors(ctx_.get()));
        generation_seen_ = ctx_.store_generation();
    }

    // device slot holding `v`, uploading it when it is not resident. Returns -1 with *rc set when the upload fails.
    int slot_for(const DescriptorView& v, uint64_t tick, int* rc) {
        const Key key{v.data, v.rows, v.dim};
        const uint64_t fp = full_check_ ? full_fingerprint(v) : fingerprint(v);
        auto it = cache_.find(key);
        if (it != cache_.end() && it->second.fingerprint == fp) {
            it->second.last_use = tick;
            ++hits_;
            return it->second.slot;
        }
        int slot;
        if (it != cache_.end()) {
            slot = it->second.slot;  // same buffer, new values: re-upload in place
        } else if (!free_slots_.empty()) {
            slot = free_slots_.back();
            free_slots_.pop_back();
        } else if (next_slot_ < capacity_) {
            slot = next_slot_++;
        } else {  // evict the least recently used frame that this batch does not use
            auto victim = cache_.end();
            for (auto c = cache_.begin(); c != cache_.end(); ++c)
                if (c->second.last_use != tick && (victim == cache_.end() || c->second.last_use < victim->second.last_use)) victim = c;
            if (victim == cache_.end()) {
                *rc = EACHAM_ERR_CAPACITY;
                return -1;
            }
            slot = victim->second.slot;
            cache_.erase(victim);
        }
        *rc = f32_ ? eacham_upload_descriptors_f32(ctx_.get(), slot, v.data, v.rows, v.dim)
                   : eacham_upload_descriptors(ctx_.get(), slot, v.data, v.rows, v.dim);
        ++uploads_;
        if (*rc != EACHAM_OK) {
            if (it != cache_.end()) cache_.erase(it);
            free_slots_.push_back(slot);
            return -1;
        }
        cache_[key] = Entry{slot, fp, tick};
        return slot;
    }

    // Only one thread at a time runs this (the leader): the cache and the context's store are its own.
    void run_batch(const std::vector<Request*>& batch) {
        if (generation_seen_ != ctx_.store_generation()) drop_cache(true);  // somebody rewrote the store (MatchAllPairs)
        hits_ = uploads_ = 0;
        std::vector<int32_t> pairs;
        std::vector<Request*> live;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const uint64_t tick = ++tick_;
            pairs.clear();
            live.clear();
            bool switch_to_f32 = false;
            for (Request* r : batch) {
                if (r->error) continue;
                int rc = EACHAM_OK;
                const int a = slot_for(r->d1, tick, &rc);
                const int b = a >= 0 ? slot_for(r->d2, tick, &rc) : -1;
                if (a < 0 || b < 0) {
                    if (!f32_ && (rc == EACHAM_ERR_NOT_INTEGER || rc == EACHAM_ERR_UNSUPPORTED)) {
                        switch_to_f32 = true;  // SuperPoint / LightGlue style floats (or a dim the int8 path lacks)
                        break;
                    }
                    r->error = std::make_exception_ptr(std::runtime_error(std::string("eacham_hip: ") + eacham_last_error(ctx_.get())));
                    continue;
                }
                pairs.push_back(a);
                pairs.push_back(b);
                live.push_back(r);
            }
            if (!switch_to_f32) break;
            f32_ = true;  // all resident frames must be of one kind: start over on the fp32 path
            drop_cache(true);
        }
        int64_t cap = 0;
        for (Request* r : live) cap += r->d1.rows > 0 ? r->d1.rows : 0;
        if (!live.empty()) {
            std::vector<int32_t> counts(live.size());
            std::vector<int64_t> offsets(live.size() + 1);
            std::vector<uint32_t> q((size_t)(cap > 0 ? cap : 1)), t(q.size());
            int64_t total = 0;
            ctx_.check(eacham_match_pairs_directed(ctx_.get(), pairs.data(), (int)live.size(), ratio_, counts.data(), offsets.data(),
                                                   q.data(), t.data(), cap, &total));
            for (size_t p = 0; p < live.size(); ++p) {
                MatchType& out = live[p]->result;
                out.reserve((size_t)counts[p]);
                for (int64_t k = offsets[p]; k < offsets[p + 1]; ++k) out.insert({q[k], t[k]});
            }
        }
        std::lock_guard<std::mutex> lk(mu_);
        stats_.calls += batch.size();
        stats_.batches += 1;
        stats_.uploads += uploads_;
        stats_.cache_hits += hits_;
    }

    float inliersRatio_;
    double ratio_;
    Context ctx_;
    const int capacity_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Request*> queue_;
    bool leader_active_ = false;
    Stats stats_;
    // owned by the leader
    std::unordered_map<Key, Entry, KeyHash> cache_;
    std::vector<int> free_slots_;
    int next_slot_ = 0;
    bool f32_ = false;
    bool full_check_ = false;  // SetFullContentCheck
    uint64_t tick_ = 0, hits_ = 0, uploads_ = 0;
    unsigned generation_seen_ = 0;
};

// The pair loop of apps/sfm/main.cpp:84-147 as one call: frames are uploaded once, every unordered
// pair is matched in both directions with the mutual check and the literal thresholds 30 / 30.
struct MatchGraph {
    std::vector<int32_t> counts;   // per pair: |mutual| if the pair became an edge, else 0
    std::vector<int64_t> offsets;  // CSR
    std::vector<uint32_t> q, t;    // Graph::Connect(n1, n2, {q -> t}); the reverse edge is the inverse map
};

// Rewrites the context's descriptor store (frame f -> id f). Integer-valued descriptors take the exact int8 path,
// anything else the fp32 MFMA path, like FeatureMatcherHip::Match.
inline MatchGraph MatchAllPairs(Context& ctx, const std::vector<DescriptorView>& frames,
                                const std::vector<std::pair<unsigned, unsigned>>& pairs, double ratio = 0.8,
                                int min_directed = 30, int min_mutual = 30) {
    ctx.store_rewritten();
    ctx.check(eacham_clear_descriptors(ctx.get()));
    int rc = EACHAM_OK;
    for (size_t f = 0; f < frames.size() && rc == EACHAM_OK; ++f)
        rc = eacham_upload_descriptors(ctx.get(), (int)f, frames[f].data, frames[f].rows, frames[f].dim);
    if (rc == EACHAM_ERR_NOT_INTEGER || rc == EACHAM_ERR_UNSUPPORTED) {
        ctx.check(eacham_clear_descriptors(ctx.get()));
        for (size_t f = 0; f < frames.size(); ++f)
            ctx.check(eacham_upload_descriptors_f32(ctx.get(), (int)f, frames[f].data, frames[f].rows, frames[f].dim));
    } else {
        ctx.check(rc);
    }
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

// The resident form for the incremental loop (apps/sfm/main.cpp:188-214 asks after every frame it adds): the graph is uploaded once,
// SetFrame refreshes the frames the loop has just changed — the one it posed and triangulated and that frame's factor neighbours —
// and Query costs two small kernels (eacham_graph_create / _set_frame / _query).
class ResidentMatchGraph {
public:
    ResidentMatchGraph(Context& ctx, const std::vector<std::pair<unsigned, unsigned>>& pairs, const MatchGraph& g,
                       const std::vector<size_t>& keypointsPerFrame)
        : ctx_(ctx) {
        std::vector<int32_t> flat(2 * pairs.size());
        for (size_t p = 0; p < pairs.size(); ++p) flat[2 * p] = (int32_t)pairs[p].first, flat[2 * p + 1] = (int32_t)pairs[p].second;
        std::vector<int64_t> kpo(keypointsPerFrame.size() + 1, 0);
        for (size_t f = 0; f < keypointsPerFrame.size(); ++f) kpo[f + 1] = kpo[f] + (int64_t)keypointsPerFrame[f];
        ctx.check(eacham_graph_create(ctx.get(), (int)keypointsPerFrame.size(), flat.data(), (int)pairs.size(), g.counts.data(), g.offsets.data(),
                                      g.q.data(), g.t.data(), kpo.data(), &h_));
    }
    ~ResidentMatchGraph() { eacham_graph_destroy(h_); }
    ResidentMatchGraph(const ResidentMatchGraph&) = delete;
    ResidentMatchGraph& operator=(const ResidentMatchGraph&) = delete;
    void SetFrame(unsigned frame, bool valid, const std::vector<uint8_t>& has3d) {
        ctx_.check(eacham_graph_set_frame(h_, (int)frame, valid ? 1 : 0, has3d.data(), (int)has3d.size()));
    }
    // several frames at once: {frame, valid, flags} — one copy and one kernel (what the loop calls after every frame it adds)
    struct FrameState { unsigned frame; bool valid; std::vector<uint8_t> has3d; };
    void SetFrames(const std::vector<FrameState>& states) {
        std::vector<int32_t> frames;
        std::vector<uint8_t> valid, flags;
        std::vector<int64_t> off{0};
        for (const auto& st : states) {
            frames.push_back((int32_t)st.frame);
            valid.push_back(st.valid ? 1 : 0);
            flags.insert(flags.end(), st.has3d.begin(), st.has3d.end());
            off.push_back((int64_t)flags.size());
        }
        ctx_.check(eacham_graph_set_frames(h_, (int)frames.size(), frames.data(), valid.data(), flags.data(), off.data()));
    }
    template <class Set>
    BestPair Query(const Set& excluded) {
        std::vector<int32_t> ex(excluded.begin(), excluded.end());
        uint32_t best[3] = {0, 0, 0};
        ctx_.check(eacham_graph_query(h_, ex.data(), (int)ex.size(), best));
        BestPair r;
        r.id = best[0], r.id2 = best[1], r.points3dCount = best[2];
        return r;
    }
private:
    Context& ctx_;
    eacham_graph* h_ = nullptr;
};

}  // namespace hip
}  // namespace eacham
