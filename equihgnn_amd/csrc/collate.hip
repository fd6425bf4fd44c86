// Host-side batch assembly (no device code): the molecules `idx` of a structure-of-arrays dataset (batch.MolStore) written as
// ONE batch into caller-owned (pinned, packed) staging arrays -- the HData.__inc__ offsets of data/utils.py:172-178 and
// PyG's Batch.from_data_list (main.py:227-229's DataLoader collate), plus the padding of batch.pad_batch.  A molecule's rows
// are contiguous in the dataset, so a batch is a few memcpy's per molecule and two passes of index arithmetic: ~25 us for a
// 256-molecule QM9 batch against ~1.2 ms for the numpy gathers it replaces (eight ranks' loader threads share a node's
// cores: tools/host_collate_ranks.py).
#include <cstring>

#include "common.h"

extern "C" int hb_collate(const HbCollate* q) {
    if (!q || q->B < 0 || (q->B > 0 && !q->idx) || !q->out_counts) return EQH_ERR_ARG;
    if (q->B > 0 && (!q->node_off || !q->he_off || !q->inc_off)) return EQH_ERR_ARG;
    const int64_t B = q->B;
    int64_t N = 0, M = 0, Z = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t m = q->idx[b];
        if (m < 0 || m >= q->n_mols) return EQH_ERR_RANGE;
        N += q->node_off[m + 1] - q->node_off[m];
        M += q->he_off[m + 1] - q->he_off[m];
        Z += q->inc_off[m + 1] - q->inc_off[m];
    }
    q->out_counts[0] = N; q->out_counts[1] = M; q->out_counts[2] = Z;
    const int64_t PN = q->padded ? q->PN : N, PM = q->padded ? q->PM : M, PZ = q->padded ? q->PZ : Z;
    // the padding molecule needs a node and a hyperedge of its own: extents that do not fit are a RANGE error (the Python
    // wrapper's "pad_to must exceed the batch"), distinct from a missing pointer
    if (q->padded && (PN <= N || PM <= M || PZ < Z)) return EQH_ERR_RANGE;
    // an array may be NULL exactly when its extent is zero (a zero-size torch tensor's data_ptr() is 0: an unpadded batch of
    // molecules without incidences, or of no molecules at all)
    const int64_t PB = B + (q->padded ? 1 : 0);
    const struct { const void* p; int64_t extent; } need[] = {
        {q->x, N}, {q->pos, N}, {q->v, Z}, {q->e, Z}, {q->edge_attr, M}, {q->e_order, M}, {q->y, B},
        {q->out_x, PN}, {q->out_pos, PN}, {q->out_batch, PN}, {q->out_edge_index0, PZ}, {q->out_edge_index1, PZ},
        {q->out_edge_attr, PM}, {q->out_e_order, PM}, {q->out_n_e, PB}, {q->out_y, PB}};
    for (const auto& a : need)
        if (!a.p && a.extent > 0) return EQH_ERR_ARG;
    int64_t n0 = 0, h0 = 0, z0 = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t m = q->idx[b];
        const int64_t ns = q->node_off[m], n = q->node_off[m + 1] - ns;
        const int64_t hs = q->he_off[m], h = q->he_off[m + 1] - hs;
        const int64_t zs = q->inc_off[m], z = q->inc_off[m + 1] - zs;
        if (n) {
            std::memcpy(q->out_x + n0 * 9, q->x + ns * 9, (size_t)n * 9 * sizeof(int64_t));
            std::memcpy(q->out_pos + n0 * 3, q->pos + ns * 3, (size_t)n * 3 * sizeof(float));
        }
        for (int64_t i = 0; i < n; ++i) q->out_batch[n0 + i] = b;
        for (int64_t i = 0; i < z; ++i) {
            q->out_edge_index0[z0 + i] = q->v[zs + i] + n0;
            q->out_edge_index1[z0 + i] = q->e[zs + i] + h0;
        }
        if (h) {
            std::memcpy(q->out_edge_attr + h0, q->edge_attr + hs, (size_t)h * sizeof(int64_t));
            std::memcpy(q->out_e_order + h0, q->e_order + hs, (size_t)h * sizeof(int64_t));
        }
        q->out_n_e[b] = h;
        q->out_y[b] = q->y[m];
        n0 += n; h0 += h; z0 += z;
    }
    if (q->padded) {
        // one dummy molecule owns every padded node and hyperedge: atoms 10 A apart on a line 10^4 A away, null incidences
        std::memset(q->out_x + N * 9, 0, (size_t)(PN - N) * 9 * sizeof(int64_t));
        for (int64_t i = 0; i < PN - N; ++i) {
            float* p = q->out_pos + (N + i) * 3;
            p[0] = 1.0e4f + 10.0f * (float)i;
            p[1] = 0.f;
            p[2] = 0.f;
            q->out_batch[N + i] = B;
        }
        for (int64_t i = Z; i < PZ; ++i) { q->out_edge_index0[i] = -1; q->out_edge_index1[i] = -1; }
        std::memset(q->out_edge_attr + M, 0, (size_t)(PM - M) * sizeof(int64_t));
        std::memset(q->out_e_order + M, 0, (size_t)(PM - M) * sizeof(int64_t));
        q->out_n_e[B] = PM - M;
        q->out_y[B] = 0.f;
    }
    return EQH_OK;
}
