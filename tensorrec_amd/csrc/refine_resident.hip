// tensorrec_amd/csrc/refine_resident.hip -- the cascade's refining launches with the ITEMS resident (round 6).
//
// What it computes is what blockmax_bf16x16_kernel<.., GRP, .., LIST> (score_blockmax.hip) computes for the fixed-capacity lists of
// trec_topk_rows_collect / trec_topk_prerefine_rows: for every kept (superblock s, user u) pair the bf16-path maximum of the
// superblock's 512 scores (tensorrec/prediction_graphs.py:50 + recommendation_graphs.py:41, reduced towards the first tf.nn.top_k of
// recommendation_graphs.py:80) written over the table entry, and every item whose bf16-path score reaches cand_floor[u] appended
// to u's candidate list.  Same MFMA (v_mfma_f32_16x16x32_bf16, chains of KT / 32, the item bias as the initial accumulator), same
// proven bound, same queue / flush / exact floor test.
//
// What differs is which operand stays.  The older kernel keeps 512 USERS in registers and streams the superblock's 512 items: a
// workgroup's whole life is one 512 x 512 product, and it starts with a chain of dependent gathers (row ids -> 128 KB of user rows,
// floors, counters, biases) and ends with a queue flush -- 15 of its ~40 us (profiles/r04_refine_clocks.json), which two workgroups
// per CU only partly hide: 0.32 of the bf16 peak.  Here a workgroup keeps the superblock's 512 ITEMS in registers (128 per wave,
// read once, contiguous) and streams a SEGMENT of the superblock's user list -- up to 2,048 users, tiles of 64 -- through LDS: the
// gathers of tile t + 1 (row ids two tiles ahead, rows by global_load_lds with one address per lane, floors / biases by wave 0) are
// in flight while tile t is multiplied, the prologue is paid once per 32 tiles, and the per-user constants live in a ring of 8 tiles
// so that queue entries stay valid for four tiles and are flushed 64 at a time.
//
// Layout of an accumulator (A = item fragment from registers, B = user fragment from LDS): lane (g = lane >> 4, lu = lane & 15)
// holds items 16 ib + 4 g .. + 3 (rows) of user 16 ub + lu (column) -- four items of ONE user, as in the older kernel, so the hit
// test (max of four >= the user's accumulator threshold), the queue entry (four scores + a code) and the flush are the same.
#include "score_common.hpp"
#include <math.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int UT = 64;                    // users per streamed tile
constexpr int SB = 512;                   // items per superblock = 4 waves x 128
constexpr int NIB = 8;                    // 16-item blocks per wave
constexpr int RING = 8;                   // tiles whose per-user constants are kept
constexpr int RQ_CAP = 320;               // queue entries per wave (20 bytes each)
constexpr int RQ_FLUSH = 64;              // a queue fuller than this is emptied before the next four hit tests (64 + 4 * 64 <= 320)

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a step loop whose index is a compile-time constant in every
// copy (left to `#pragma unroll` the step loop below stayed a loop over the user blocks, and the rotating fragment registers
// uf[(st + 2) % 3] became 376 v_cndmask per step)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct ResidentParams {
    const void* U;                        // bf16 [n_users][KT]
    const void* T;                        // bf16 [n_items][KT]
    int64_t n_items;
    const float* u_bias;                  // nullable
    const float* t_bias;                  // nullable
    const int32_t* row_count;             // [n_sb] users listed per superblock (may exceed rcap: clamped)
    const int32_t* row_user;              // [n_sb][rcap]
    int32_t rcap, n_sb;
    float* blockmax;
    int64_t bm_stride;
    const float* cand_floor;
    int32_t* cand_n;
    int2* cand;
    int32_t cand_cap, t_index_base;
    const int32_t* wg_map;                // nullable [n_wgs]: slot = s * segs_per_row + j
    int32_t segs_per_row, seg_rows;
    int32_t diag;                         // tuning refine_resident_diag (timing experiments only, WRONG results): 1 = contiguous user rows instead of
                                          // the gather, 2 = nothing is ever listed, 8 = no per-user constant gathers
};

template <int KT, bool BIAS>
__global__ __launch_bounds__(256, 2) void refine_resident_kernel(ResidentParams p)
{
    constexpr int RB = KT * 2;               // bytes per operand row
    constexpr int CH = RB / 16;              // 16-byte chunks per row
    constexpr int KS = KT / 32;              // MFMA k-steps per block
    constexpr int TILE_BYTES = UT * RB;
    constexpr int NSLOT = UT * CH / 256;
    constexpr int NSTEP = (UT / 16) * KS;
    static_assert(KT == 64 || KT == 128, "K = 64 / 128");
    static_assert((256 / CH) % CH == 0, "swizzle term independent of the slot round");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ibias = (float*)(smem + 2 * TILE_BYTES);                       // [512]
    f32x4* qv_all = (f32x4*)(ibias + SB);                                 // [4][RQ_CAP]
    int32_t* qc_all = (int32_t*)(qv_all + 4 * RQ_CAP);                    // [4][RQ_CAP]
    float* q_fl = (float*)(qc_all + 4 * RQ_CAP);                          // [RING * 64] exact floor of the user
    float* q_bu = q_fl + RING * UT;                                       //             user bias
    int32_t* q_id = (int32_t*)(q_bu + RING * UT);                         //             user row (-1: padding)
    float* q_ta = (float*)(q_id + RING * UT);                             //             accumulator threshold
    float* wmax = q_ta + RING * UT;                                       // [2][4][64] per-wave maxima of a tile's users

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lu = lane & 15;
    f32x4* qv = qv_all + wave * RQ_CAP;
    int32_t* qc = qc_all + wave * RQ_CAP;

    const int slot = p.wg_map ? p.wg_map[blockIdx.x] : (int)blockIdx.x;
    const int s = slot / p.segs_per_row;
    if (s >= p.n_sb) return;                                             // idle entry of the map
    int cnt = p.row_count[s];
    if (cnt > p.rcap) cnt = p.rcap;
    const int first = (slot - s * p.segs_per_row) * p.seg_rows;
    int n_here = cnt - first;
    if (n_here > p.seg_rows) n_here = p.seg_rows;
    if (n_here <= 0) return;
    const int ntiles = (n_here + UT - 1) / UT;
    const int32_t* list = p.row_user + (int64_t)s * p.rcap + first;
    const int64_t item0 = (int64_t)s * SB;

    // ---- wave 0: the row ids of tiles 0 and 1, the constants of tile 0 (two dependent round trips; the item rows below ride along)
    auto user_consts = [&](int32_t uid, float flv, int32_t cnv, float buv, int ring_slot) __attribute__((always_inline)) {
        const bool rl = uid >= 0;
        float fl = rl ? flv : INFINITY;
        if (rl && cnv > p.cand_cap) fl = INFINITY;                       // the list is already incomplete: the user will be re-done
        const float bu = (BIAS && p.u_bias && rl) ? buv : 0.f;
        // acc + bu >= fl (evaluated exactly when the queue is emptied) implies acc >= ta: fl - bu less three roundings
        const float t = fl - bu;
        float ta = t - (fabsf(fl) + fabsf(bu) + fabsf(t)) * 2.4e-7f;
        if (bu == 0.f) ta = fl;
        if (!(fl < INFINITY) || (p.diag & 2)) ta = INFINITY;             // padding rows, users without a usable bound: nothing is listed
        else if (!(ta == ta)) ta = -INFINITY;                            // (fl = -inf: everything is)
        q_fl[ring_slot * UT + lane] = fl;
        q_bu[ring_slot * UT + lane] = bu;
        q_ta[ring_slot * UT + lane] = ta;
    };
    const float* ubias_or_floor = (BIAS && p.u_bias) ? p.u_bias : p.cand_floor;
    if (wave == 0) {
        const int32_t a0 = list[lane < n_here ? lane : n_here - 1];
        const int32_t a1 = list[UT + lane < n_here ? UT + lane : n_here - 1];
        const int32_t i0 = lane < n_here ? a0 : -1, i1 = UT + lane < n_here ? a1 : -1;
        q_id[0 * UT + lane] = i0;
        q_id[1 * UT + lane] = i1;
        const int64_t r = i0 >= 0 ? i0 : 0;
        const float flv = p.cand_floor[r];
        const int32_t cnv = p.cand_n[r];
        const float buv = ubias_or_floor[r];
        user_consts(i0, flv, cnv, buv, 0);
    }
    // ---- resident item fragments: lane holds k = 32 ks + 8 g + 0..7 of item 16 ib + lu of its wave's 128 items ----
    bf16x8 ifr[NIB][KS];
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib) {
        int64_t row = item0 + wave * 128 + ib * 16 + lu;
        if (row >= p.n_items) row = p.n_items - 1;                       // duplicate of the last valid item: maxima unchanged, never listed
        const char* src = (const char*)p.T + row * (int64_t)RB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ifr[ib][ks] = *(const bf16x8*)(src + (ks * 4 + g) * 16);
    }
    {
#pragma unroll
        for (int h = 0; h < SB / 256; ++h) {
            int64_t it = item0 + h * 256 + tid;
            if (it >= p.n_items) it = p.n_items - 1;
            ibias[h * 256 + tid] = (BIAS && p.t_bias) ? p.t_bias[it] : 0.f;
        }
    }
    __syncthreads();

    auto stage_issue = [&](int tile, int buf) __attribute__((always_inline)) {
        const int32_t* ids = q_id + (tile & (RING - 1)) * UT;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int q = i * 256 + tid;
            const int row = q / CH, chp = q % CH;
            const int32_t uid = ids[row];
            int64_t r = uid >= 0 ? uid : 0;                              // padding rows compute on user 0 (never listed, never stored)
            if (p.diag & 1) r = first + tile * UT + row;
            const char* src = (const char*)p.U + r * (int64_t)RB + ((chp ^ (row & (CH - 1))) * 16);
            char* dst = smem + buf * TILE_BYTES + (i * 256 + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = lu * RB + (((ks * 4 + g) ^ (lu & (CH - 1))) * 16);      // rows lu + 16 b: same swizzle

    int qn = 0;                                                          // wave-uniform: entries in this wave's queue
    // One lane, one entry: the four scores of items 4 c .. 4 c + 3 of the wave's 128 for one user of the ring; a score that reaches
    // the user's floor with the user bias added takes the next slot of the user's list.
    auto queue_entry = [&](int i, int n) __attribute__((always_inline)) {
        const bool in = i < n;
        const int32_t code = in ? qc[i] : 0;
        const f32x4 a = in ? qv[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ul = code >> 16;
        const int32_t uid = in ? q_id[ul] : -1;
        const float fl = q_fl[ul], bu = q_bu[ul];
        const int64_t it0 = item0 + wave * 128 + (int64_t)(code & 0xffff) * 4;
        float v[4];
        int32_t sl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = BIAS ? a[e] + bu : a[e];
            sl[e] = 0x7fffffff;
            if (uid >= 0 && v[e] >= fl && it0 + e < p.n_items) sl[e] = atomicAdd(p.cand_n + uid, 1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (sl[e] < p.cand_cap)
                p.cand[(int64_t)uid * p.cand_cap + sl[e]] = make_int2((int32_t)(it0 + e) + p.t_index_base, __float_as_int(v[e]));
    };
    auto queue_flush = [&](auto unrc) __attribute__((always_inline)) {
        constexpr int UNR = decltype(unrc)::value;
        const int n = qn;
#pragma unroll 1
        for (int i0 = 0; i0 < n; i0 += 64 * UNR) {
#pragma unroll
            for (int j = 0; j < UNR; ++j) queue_entry(i0 + j * 64 + lane, n);
        }
        qn = 0;
    };

    f32x4 acc[NIB];
    // One tile = four 16-user blocks x eight 16-item blocks.  The hit tests of an accumulator need its finished value, so a wave
    // that tests right behind its MFMAs leaves the matrix pipe idle meanwhile (the first version: 0.37 of the peak with nothing ever
    // listed).  The eight accumulators are therefore two GROUPS of four: while group A of user block ub is multiplied (KS steps of
    // four MFMAs) group B of block ub - 1 is tested, one accumulator per step behind its step's MFMAs; while group B of ub is
    // multiplied, group A of ub is tested.  Group B of the last block is tested after the loop.
    auto tile_body = [&](int buf, int t) __attribute__((always_inline)) {
        constexpr int HG = NIB / 2;                                      // accumulators per group
        constexpr int TPK = HG / KS;                                     // hit tests per k-step (1 at K = 128, 2 at K = 64)
        constexpr int NST = (UT / 16) * 2 * KS;                          // steps per tile: (user block, group, k-step)
        const int boff = buf * TILE_BYTES;
        const float* ibw = ibias + wave * 128 + 4 * g;                  // the item biases of result rows 4 g .. 4 g + 3 of block ib: + 16 ib
        const int rs = (t & (RING - 1)) * UT;
        float* wm = wmax + ((t & 1) * 4 + wave) * UT;
        const char* kb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kb[ks] = smem + boff + koff[ks];
        auto hit_test = [&](const f32x4& a, float thr, float& bmx, int ubi, int ibi) __attribute__((always_inline)) {
            const float m4 = fmaxf(fmaxf(fmaxf(a[0], a[1]), a[2]), a[3]);
            bmx = fmaxf(bmx, m4);
            const bool hit = m4 >= thr;
            const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
            if (hm != 0ull) {                                            // wave-uniform
                const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(hm >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((unsigned int)hm, 0u));
                if (hit) {
                    int lo = lane;
                    asm volatile("" : "+v"(lo));                         // (the codes are computed in the hit path, not hoisted and spilled)
                    qv[pos] = a;
                    qc[pos] = ((rs + ubi * 16 + (lo & 15)) << 16) | ((ibi << 2) + (lo >> 4));
                }
                qn += __builtin_popcountll(hm);
            }
        };
        // a block's 16 users: the four row groups' maxima meet, row group 0 leaves the wave's maximum for the combine
        auto finish_block = [&](float bm, int ubi) __attribute__((always_inline)) {
            bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            if (g == 0) wm[ubi * 16 + lu] = bm;
        };
        bf16x8 uf[3];
        uf[0] = *(const bf16x8*)(kb[0]);
        uf[1] = *(const bf16x8*)(kb[1 % KS] + ((1 / KS) / 2) * 16 * RB);
        // the item biases (initial accumulators) of the NEXT phase's four blocks are read two steps ahead, like the fragments: read
        // right in front of their MFMAs, every phase began with an exposed LDS round trip (lgkmcnt(0) also waits for the prefetch)
        f32x4 c0n[HG];
#pragma unroll
        for (int i = 0; i < HG; ++i) {
            c0n[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (BIAS) c0n[i] = *(const f32x4*)(ibw + 16 * i);
        }
        float thr_cur = q_ta[rs + lu], thr_prev = INFINITY;
        float bm_cur = -INFINITY, bm_prev = -INFINITY;
        __builtin_amdgcn_sched_barrier(0);
        static_for<NST>([&](auto stc) __attribute__((always_inline)) {
            constexpr int st = decltype(stc)::value;
            constexpr int ph = st / KS, ks = st % KS;                    // phase = (user block, group)
            constexpr int ub = ph / 2, grp = ph % 2;
            if (st + 2 < NST) {
                constexpr int ph2 = (st + 2) / KS, ks2 = (st + 2) % KS;
                uf[(st + 2) % 3] = *(const bf16x8*)(kb[ks2] + (ph2 / 2) * 16 * RB);
            }
            if (ks == 0 && __builtin_expect(qn > RQ_FLUSH, 0)) queue_flush(std::integral_constant<int, 1>{});
            if (ks == 0 && grp == 0 && ub > 0) thr_cur = q_ta[rs + ub * 16 + lu];
#pragma unroll
            for (int i = 0; i < HG; ++i) {
                const int ib = grp * HG + i;
                if (ks == 0) {
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ifr[ib][0], uf[st % 3], c0n[i], 0, 0, 0);
                } else {
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ifr[ib][ks], uf[st % 3], acc[ib], 0, 0, 0);
                }
            }
            if (BIAS && ks == (KS >= 2 ? KS - 2 : 0) && st + 2 < NST) {
                // (behind the MFMAs that consumed c0n at ks == 0 -- for KS == 2 this IS step 0 -- the next phase's biases)
                constexpr int gn = (ph + 1) % 2;
#pragma unroll
                for (int i = 0; i < HG; ++i) c0n[i] = *(const f32x4*)(ibw + 16 * (gn * HG + i));
            }
            if (grp == 0) {
                if (ub > 0) {
#pragma unroll
                    for (int j = 0; j < TPK; ++j) hit_test(acc[HG + ks * TPK + j], thr_prev, bm_prev, ub - 1, HG + ks * TPK + j);
                    if (ks == KS - 1) finish_block(bm_prev, ub - 1);
                }
            } else {
#pragma unroll
                for (int j = 0; j < TPK; ++j) hit_test(acc[ks * TPK + j], thr_cur, bm_cur, ub, ks * TPK + j);
                if (ks == KS - 1) { thr_prev = thr_cur; bm_prev = bm_cur; bm_cur = -INFINITY; }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int j = 0; j < HG; ++j) hit_test(acc[HG + j], thr_prev, bm_prev, UT / 16 - 1, HG + j);
        finish_block(bm_prev, UT / 16 - 1);
    };
    // the maxima of tile T's users over the four waves' items, + user bias, over the table entries (wave 1, one user per lane)
    auto combine = [&](int T) __attribute__((always_inline)) {
        const int rs = (T & (RING - 1)) * UT;
        const float* wm = wmax + (T & 1) * 4 * UT;
        float v = fmaxf(fmaxf(wm[lane], wm[UT + lane]), fmaxf(wm[2 * UT + lane], wm[3 * UT + lane]));
        const int32_t uid = q_id[rs + lane];
        if (BIAS) v = v + q_bu[rs + lane];
        if (uid >= 0) p.blockmax[(int64_t)s * p.bm_stride + uid] = v;
    };

    stage_issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        // wave 0: row ids of tile t + 2, constants of tile t + 1 (its ids were written one iteration ago)
        int32_t uidn = -1, uid1 = -1, cnv = 0;
        float flv = 0.f, buv = 0.f;
        bool vn = false;
        if (wave == 0) {
            const int idx = (t + 2) * UT + lane;
            vn = idx < n_here;
            uidn = list[vn ? idx : n_here - 1];
            uid1 = q_id[((t + 1) & (RING - 1)) * UT + lane];
            const int64_t r = (uid1 >= 0 && !(p.diag & 8)) ? uid1 : 0;
            flv = p.cand_floor[r];
            cnv = p.cand_n[r];
            buv = ubias_or_floor[r];
        }
        if (t + 1 < ntiles) stage_issue(t + 1, buf ^ 1);
        if (wave == 1 && t > 0) combine(t - 1);
        tile_body(buf, t);
        if (wave == 0) {
            // (pinned: the select on uidn had been hoisted into the block that issues the load, with an s_waitcnt in front of wave 0's
            // share of the row DMA -- one global round trip per tile in front of every tile's barrier)
            asm volatile("" : "+v"(uidn), "+v"(flv), "+v"(cnv), "+v"(buv));
            q_id[((t + 2) & (RING - 1)) * UT + lane] = vn ? uidn : -1;
            user_consts(uid1, flv, cnv, buv, (t + 1) & (RING - 1));
        }
        if ((t & 3) == 3 || t + 1 == ntiles) queue_flush(std::integral_constant<int, 2>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (wave == 1) combine(ntiles - 1);
}

template <int KT, bool BIAS>
int launch_resident(const ResidentParams& p, unsigned blocks, hipStream_t st)
{
    constexpr int LDS = 2 * UT * KT * 2 + SB * 4 + 4 * RQ_CAP * 20 + 4 * RING * UT * 4 + 2 * 4 * UT * 4;
    auto kern = refine_resident_kernel<KT, BIAS>;
    static bool attr_set = false;
    if (!attr_set && LDS > 32 * 1024) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS, st, p);
    return trec_check_launch("trec_score_gemm_refine_candidates_resident");
}

}  // namespace

// The refining launch over the fixed-capacity lists row_user [n_sb][rcap] (row_count [n_sb] users each, clamped to rcap; 0 for
// the hot superblocks) with the superblock's items resident: workgroup w takes segment wg_map[w] % segs_per_row (seg_rows users, a
// multiple of 64) of superblock wg_map[w] / segs_per_row; entries >= n_sb * segs_per_row are idle (trec_topk_rows_wg_map_ex with
// group_rows = seg_rows).  sb_rows must be 512.  Everything else as trec_score_gemm_refine_candidates.
extern "C" int trec_score_gemm_refine_candidates_resident(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_items,
                                                          const float* user_bias, const float* item_bias, int32_t sb_rows, int32_t n_sb,
                                                          const int32_t* row_count, const int32_t* row_user, int32_t rcap,
                                                          float* blockmax, int64_t bm_stride, const float* cand_floor, int32_t* cand_n,
                                                          void* cand, int32_t cand_cap, int32_t item_index_base, const int32_t* wg_map,
                                                          int32_t n_wgs, int32_t segs_per_row, int32_t seg_rows, void* stream)
{
    TREC_REQUIRE(users_bf16 && items_bf16 && row_count && row_user && blockmax && cand_floor && cand_n && cand && wg_map,
                 "trec_score_gemm_refine_candidates_resident: null pointer");
    TREC_REQUIRE(kpad == 64 || kpad == 128, "trec_score_gemm_refine_candidates_resident: kpad must be 64 or 128");
    TREC_REQUIRE(sb_rows == SB, "trec_score_gemm_refine_candidates_resident: sb_rows must be 512");
    TREC_REQUIRE(n_sb >= 1 && rcap >= 1 && cand_cap >= 1 && n_items >= 1 && n_wgs >= 0 && segs_per_row >= 1 && seg_rows >= UT &&
                 seg_rows % UT == 0 && (int64_t)segs_per_row * seg_rows >= rcap,
                 "trec_score_gemm_refine_candidates_resident: bad sizes");
    if (n_wgs == 0) return TREC_OK;
    ResidentParams p = {};
    p.U = users_bf16; p.T = items_bf16; p.n_items = n_items; p.u_bias = user_bias; p.t_bias = item_bias;
    p.row_count = row_count; p.row_user = row_user; p.rcap = rcap; p.n_sb = n_sb;
    p.blockmax = blockmax; p.bm_stride = bm_stride;
    p.cand_floor = cand_floor; p.cand_n = cand_n; p.cand = (int2*)cand; p.cand_cap = cand_cap; p.t_index_base = item_index_base;
    p.wg_map = wg_map; p.segs_per_row = segs_per_row; p.seg_rows = seg_rows; p.diag = trec_get_tuning("refine_resident_diag", 0);
    hipStream_t st = (hipStream_t)stream;
    const bool bias = user_bias || item_bias;
    if (kpad == 128) return bias ? launch_resident<128, true>(p, (unsigned)n_wgs, st) : launch_resident<128, false>(p, (unsigned)n_wgs, st);
    return bias ? launch_resident<64, true>(p, (unsigned)n_wgs, st) : launch_resident<64, false>(p, (unsigned)n_wgs, st);
}
