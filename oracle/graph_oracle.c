/*
 * graph_oracle.c — CPU restatement of eacham's view-graph query on the match graph.
 *
 * TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (the reference has no tests for this path).
 *   Graph::Connect, both directions      /root/reference/modules/sfm/data/Graph.h:30-41, apps/sfm/main.cpp:144-145
 *   Graph::GetBestPairForValid           /root/reference/modules/sfm/data/Graph.h:59-106
 * The match graph arrives in the CSR wire format of the matcher (pairs, counts, offsets, q, t): pair p
 * with counts[p] > 0 is the factor f1 -> f2 with matches q -> t and the factor f2 -> f1 with t -> q.
 * The reference walks `nodes` (std::map: ascending id) and each node's `factors`
 * (std::unordered_map: no defined order); here neighbours are visited in ascending id. A candidate
 * replaces the best unless `bestScore > points3dCount`, so among equal counts the LAST visited wins.
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct { uint32_t node, other; uint32_t count; } edge_t;

static int edge_cmp(const void* a, const void* b) {
    const edge_t *x = a, *y = b;
    if (x->node != y->node) return x->node < y->node ? -1 : 1;
    if (x->other != y->other) return x->other < y->other ? -1 : 1;
    return 0;
}

/* kp_has3d[kp_offsets[f] + k] = node f HasPoint3d(k) && !IsPoint3dTwoView(k).
 * edge_counts (optional, 2*npairs): points3dCount of the factor f1->f2 and of f2->f1 for every pair. */
void oracle_graph_best_pair(int n_frames, const int32_t* pairs, int npairs, const int32_t* counts, const int64_t* offsets,
                            const uint32_t* q, const uint32_t* t, const uint8_t* valid, const uint8_t* excluded,
                            const int64_t* kp_offsets, const uint8_t* kp_has3d, uint32_t* edge_counts, uint32_t* best) {
    edge_t* e = (edge_t*)malloc(sizeof(edge_t) * (size_t)(2 * npairs + 1));
    int ne = 0;
    for (int p = 0; p < npairs; ++p) {
        const int f1 = pairs[2 * p], f2 = pairs[2 * p + 1];
        uint32_t c12 = 0, c21 = 0;
        for (int64_t k = offsets[p]; k < offsets[p] + counts[p]; ++k) {
            c12 += kp_has3d[kp_offsets[f1] + q[k]];
            c21 += kp_has3d[kp_offsets[f2] + t[k]];
        }
        if (edge_counts) { edge_counts[2 * p] = c12; edge_counts[2 * p + 1] = c21; }
        if (counts[p] > 0) {
            e[ne++] = (edge_t){(uint32_t)f1, (uint32_t)f2, c12};
            e[ne++] = (edge_t){(uint32_t)f2, (uint32_t)f1, c21};
        }
    }
    qsort(e, (size_t)ne, sizeof(edge_t), edge_cmp);
    float best_score = 0;
    best[0] = best[1] = 0xffffffffu;
    best[2] = 0;
    for (int i = 0; i < ne; ++i) {
        if (!valid[e[i].node]) continue;
        if (valid[e[i].other] || (excluded && excluded[e[i].other])) continue;
        if (best_score > (float)e[i].count) continue;
        best_score = (float)e[i].count;
        best[0] = e[i].node; best[1] = e[i].other; best[2] = e[i].count;
    }
    (void)n_frames;
    free(e);
}
