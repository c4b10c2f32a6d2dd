// Flash-style multi-head self attention, head_dim = 64, bf16 in / fp32 softmax / bf16 out (gfx950).
//
// Replaces the attention the reference reaches through timm `Attention` (F.scaled_dot_product_attention),
// HF `CLIPAttention` (modeling_video.py:98,168) and HF `VivitSelfAttention` (vivit.py:104): no mask, no
// dropout, softmax(q k^T / sqrt(64)) v. Sequences: LanguageBind 257, DINOv2 261, SigLIP 196 (many short
// ones) and ViViT 3137 (one long one per video) -- the same kernel streams 64-key tiles with an online
// softmax for all of them.
//
// Layout trick (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand"):
// scores are computed transposed, S^T = K Q^T with v_mfma_f32_32x32x16_bf16, so each lane owns ONE query
// (column = lane & 31) and 16 keys per 32-key block in its accumulator registers. Row max / row sum are then
// in-lane reductions plus one cross-half exchange, the rescale factor is a per-lane scalar, and the
// exponentiated tile is already in B-operand order for O^T = V^T P^T: no LDS round trip for P.
// V^T fragments come either from a row-major V tile read with the hardware transpose load
// (ds_read_b64_tr_b16, VTR = true) or from a tile transposed while staging (VTR = false).
#include <atomic>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "prof.h"

// Diagnostic builds only (tools/probes/attn_stamps.hip defines MERV_ATTN_STAMPS before including this file): s_memtime
// stamps per wave into a buffer of their own. The product build compiles none of it.
#ifndef MERV_ATTN_ABL
#define MERV_ATTN_ABL 0  // probe builds: 1 = no key-tile loop (memory side alone), 2 = no K / V DMA (compute side alone), 3 = no output stores; results garbage
#endif
#ifdef MERV_ATTN_STAMPS
__device__ unsigned long long* g_attn_stamps = nullptr;  // [block][wave][32]
#define MERV_STAMP(k)                                                                                       \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        unsigned long long t__;                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (g_attn_stamps && (threadIdx.x & 63) == 0 && (k) < 32) {                                         \
            const int blk__ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);               \
            g_attn_stamps[((size_t)blk__ * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + (k)] = t__;      \
        }                                                                                                   \
    } while (0)
#else
#define MERV_STAMP(k) do { } while (0)
#endif

namespace merv {
namespace {

constexpr float LOG2E = 1.4426950408889634f;
// raw v_exp_f32 (2^x): exp2f() adds range scaling (v_cmp + v_cndmask + v_ldexp per call) that softmax does not need --
// arguments are <= 0 and underflow to 0 is the wanted result; -inf -> 0.
MERV_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr int HD = 64;               // head_dim
constexpr int KROW = 128;            // bytes per K row in LDS
constexpr int VROW_TR = 192;         // bytes per V row (row-major image for transposed reads; 192 keeps 4 rows on disjoint banks)
constexpr int VT_ROW = 136;          // bytes per V^T row (64 keys + 8 B pad => conflict-free ds_read_b64)

MERV_DEVICE int kswz(int key) { return (key >> 1) & 7; }  // 16 consecutive rows -> 16 distinct 16-B slots

MERV_DEVICE bf16x8 join8(s16x4 a, s16x4 b) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

MERV_DEVICE bf16x8 pack8(const f32x16& s, int base) {
    u32x4 w;
    w[0] = pack2bf(s[base + 0], s[base + 1]);
    w[1] = pack2bf(s[base + 2], s[base + 3]);
    w[2] = pack2bf(s[base + 4], s[base + 5]);
    w[3] = pack2bf(s[base + 6], s[base + 7]);
    return __builtin_bit_cast(bf16x8, w);
}

// V^T A-operand fragment for O^T[d-block db] over the 16 keys starting at key_base (tile-local).
// Element j of lane (r = lane & 31, h = lane >> 5) must be V[key_base + 8*(j>>2) + 4*h + (j&3)][db*32 + r].
// SWZ128 (resident image): V rows are 128 B, unpadded; the 64-byte halves of rows 2, 3 (mod 4) are swapped so that the four
// rows a 32-lane half reads (4 rows x 64 B) cover the 256-byte bank row exactly once (tile bases are multiples of 16 rows).
template <bool VTR, bool SWZ128 = false>
MERV_DEVICE bf16x8 load_vt_frag(const char* v_lds, int key_base, int db, int lane, int vrow_bytes) {
    const int h = lane >> 5;
    if constexpr (SWZ128) {
        const int q4 = (lane & 15) >> 2, p4 = lane & 3;
        const int half = db ^ ((q4 >> 1) & 1);
        const char* a0 = v_lds + (key_base + 4 * h + q4) * 128 + half * 64 + 32 * ((lane >> 4) & 1) + 8 * p4;
        const char* a1 = a0 + 8 * 128;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
        return join8(lo, hi);
    } else if constexpr (VTR) {
        // ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block; lane 4q+p supplies row q, cols 4p..4p+3;
        // lane i of the group receives column i of the 4 rows.
        const int q4 = (lane & 15) >> 2, p4 = lane & 3;
        const int col = db * 32 + 16 * ((lane >> 4) & 1) + 4 * p4;
        const char* a0 = v_lds + (key_base + 4 * h + q4) * vrow_bytes + col * 2;
        const char* a1 = a0 + 8 * vrow_bytes;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
        return join8(lo, hi);
    } else {
        const int r = lane & 31;
        const char* a0 = v_lds + (db * 32 + r) * vrow_bytes + (key_base + 4 * h) * 2;
        s16x4 lo = *(const s16x4*)a0;
        s16x4 hi = *(const s16x4*)(a0 + 16);
        return join8(lo, hi);
    }
}

// Write two V rows (keys 2kp, 2kp+1; 8 d-values starting at 8c) transposed into V^T[d][key].
MERV_DEVICE void write_vt_pair(char* vt_lds, int vt_row_bytes, int kp, int c, u32x4 a, u32x4 b) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t lo = (a[w] & 0xffffu) | (b[w] << 16);
        const uint32_t hi = (a[w] >> 16) | (b[w] & 0xffff0000u);
        *(uint32_t*)(vt_lds + (8 * c + 2 * w) * vt_row_bytes + 4 * kp) = lo;
        *(uint32_t*)(vt_lds + (8 * c + 2 * w + 1) * vt_row_bytes + 4 * kp) = hi;
    }
}

// NW waves per block, each wave owns QPW consecutive 32-row query tiles: one block covers NW*QPW*32 queries of one
// (sequence, head) and streams the K/V tiles ONCE for all of them (L = 257/261 -> 3 waves x 3 tiles = 288 rows, one
// block per (sequence, head); L = 196 -> 4 x 2; L = 3137 -> 13 blocks of 4 x 2).
// XQ = true (sequences of NW*QPW*32 + 1..8 tokens: 257 = 8 * 32 + 1, 261 = 8 * 32 + 5): the few query rows past the last full
// tile do not get a padded 32-row tile of their own wave (a ninth tile on a 3 x 3 block: three waves on four SIMDs, no second
// score set) -- the block stays 4 waves x 2 tiles, and the extra rows' attention is split over the waves BY KEY TILE: wave
// t % NW multiplies them against key tile t right after its own two tiles (K / V of that tile are in LDS anyway), leaves
// {partial O, max, sum} in LDS, and the partials are merged after the loop (flash-decoding style). Bit pattern of the
// result differs from the single-pass order only by fp32 summation order.
// RES = true (with XQ; L <= 264): ALL K / V rows of the (sequence, head) are brought into LDS once by LDS-DMA (source-side
// swizzle, 66 one-KiB pieces over the four waves) and the key-tile loop has no barrier, no staging registers and no global
// load: the waves drift freely (the wave that multiplies the extra rows in iteration t no longer holds the others at a
// barrier), and the second block of the CU computes while this one waits for its single load phase. 78 KB per block: two
// blocks per CU.
constexpr int XQ_ROWS = 8, XQ_SLOTS = 5;
constexpr int RES_KROWS = 264, RES_VROWS = 272;  // K rows past 264 are read (and masked) from whatever follows; V rows must be finite
// MXQ (compile time since round 5): the output goes out as MXFP8; with the choice a run-time test inside every row chunk of the output loop hipcc issued
// the chunks' LDS read-backs one by one, each waited for in front of its own store (as in the GEMM epilogue, gemm.hip WHOLE)
template <bool VTR, int NW, int QPW, bool XQ = false, bool RES = false, bool MXQ = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_kernel(AttnArgs p) {
    static_assert(!RES || (VTR && XQ && NW * QPW * 32 == 256), "resident form: 4 x 2 block with the extra-row split only");
    constexpr int NT = NW * 64;
    constexpr int V_BYTES = RES ? RES_VROWS * 128 : (VTR ? 64 * VROW_TR : 64 * VT_ROW);
    constexpr int K_BYTES = RES ? RES_KROWS * KROW : 64 * KROW;
    constexpr int VROWB = RES ? 128 : (VTR ? VROW_TR : VT_ROW);
    constexpr int KSTG = (512 + NT - 1) / NT;   // 16-byte chunks of a 64x64 bf16 tile per thread
    constexpr int PSTG = (256 + NT - 1) / NT;   // (key pair, chunk) items per thread for the transposing V path
    constexpr int OUT_BYTES = NW * 32 * 128;    // output staging (reuses the K/V region after the last tile)
    constexpr int KV_BYTES = (K_BYTES + V_BYTES) > OUT_BYTES ? (K_BYTES + V_BYTES) : OUT_BYTES;
    constexpr int XQ_BYTES = XQ ? XQ_SLOTS * XQ_ROWS * (HD + 4) * 4 : 0;  // per (slot, row): 64 partial outputs, max, two half sums
    constexpr int LDS_BYTES = KV_BYTES + XQ_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    char* k_lds = smem;
    char* v_lds = smem + K_BYTES;
    float* xq_lds = (float*)(smem + KV_BYTES);  // [slot][row][HD + 4]

    const int tid = threadIdx.x;
    __builtin_assume(tid >= 0 && tid < NT);  // (the staging loops' `idx < 512` tests fold away where NT divides 512)
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int seq = blockIdx.z, head = blockIdx.y;
    const int L = p.L, D = p.D;
    const int ld = 3 * D;
    const bf16_t* base = p.qkv + (size_t)seq * L * ld;
    const bf16_t* kbase = base + D + head * HD;
    const bf16_t* vbase = base + 2 * D + head * HD;

    const int q_base = (blockIdx.x * NW + wave) * (QPW * 32);  // first query row of this wave
    MERV_STAMP(0);

    // Q^T B-operand fragments: element j of step s = Q[q][16 s + 8 h + j]
    auto ld16 = [&](const bf16_t* ptr) { return *(const u32x4*)ptr; };
    u32x4 qraw[QPW][4], qxraw[4];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int q_row = q_base + qi * 32 + r;
        const int q_ld = q_row < L ? q_row : L - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qraw[qi][s] = ld16(base + (size_t)q_ld * ld + head * HD + 16 * s + 8 * h);
    }
    constexpr int XQ0 = NW * QPW * 32;  // first extra query row (XQ launches have one block per (sequence, head))
    if constexpr (XQ) {
        const int q_ld = XQ0 + r < L ? XQ0 + r : L - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qxraw[s] = ld16(base + (size_t)q_ld * ld + head * HD + 16 * s + 8 * h);
    }

    f32x16 oacc[QPW][2];
    float m_run[QPW], l_run[QPW];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[qi][0][i] = 0.f; oacc[qi][1][i] = 0.f; }
        m_run[qi] = -INFINITY;
        l_run[qi] = 0.f;
    }
    const float sc = p.scale * LOG2E;

    // staging registers (issue global loads early, write LDS after the barrier)
    u32x4 kreg[KSTG], vreg[VTR ? KSTG : 2 * PSTG];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < KSTG; ++i) {
            const int idx = tid + NT * i;
            if (idx < 512) {
                int key = kv0 + (idx >> 3);
                key = key < L ? key : L - 1;
                kreg[i] = *(const u32x4*)(kbase + (size_t)key * ld + (idx & 7) * 8);
                if constexpr (VTR) vreg[i] = *(const u32x4*)(vbase + (size_t)key * ld + (idx & 7) * 8);
            }
        }
        if constexpr (!VTR) {
#pragma unroll
            for (int i = 0; i < PSTG; ++i) {
                const int pidx = tid + NT * i;
                if (pidx < 256) {
                    const int kp = pidx >> 3, c = pidx & 7;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        int key = kv0 + 2 * kp + u;
                        key = key < L ? key : L - 1;
                        vreg[2 * i + u] = *(const u32x4*)(vbase + (size_t)key * ld + c * 8);
                    }
                }
            }
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < KSTG; ++i) {
            const int idx = tid + NT * i;
            if (idx < 512) {
                const int key = idx >> 3, c = idx & 7;
                *(u32x4*)(k_lds + key * KROW + ((c ^ kswz(key)) * 16)) = kreg[i];
                if constexpr (VTR) *(u32x4*)(v_lds + key * VROW_TR + c * 16) = vreg[i];
            }
        }
        if constexpr (!VTR) {
#pragma unroll
            for (int i = 0; i < PSTG; ++i) {
                const int pidx = tid + NT * i;
                if (pidx < 256) write_vt_pair(v_lds, VT_ROW, pidx >> 3, pidx & 7, vreg[2 * i], vreg[2 * i + 1]);
            }
        }
    };

    const int ntiles = (L + 63) / 64;
    if constexpr (RES) {
        // one LDS-DMA wave-instruction = 8 rows x 128 B = 1 KiB, lane-linear in LDS; the chunk swizzles of the readers are
        // applied to the per-lane SOURCE address (cdna_hip_programming.md rule 21); rows past L re-read the last row
        // Two load phases. First: key tiles 0 and 1 (next to the Q fragment loads above, whose first use makes hipcc wait
        // vmcnt(0) anyway: with a register load in flight beside LDS-DMA it drains the whole queue, cdna_hip_programming.md
        // section 5 "Three .s-level traps" (b)). Behind the barrier: the DMA of tiles 2 .. 4, which lands under the compute of
        // tiles 0 and 1 and is waited for (vmcnt(0) + barrier) at the top of tile 2 -- no register load is in flight then, so the
        // compiler inserts no wait of its own inside the loop.
        const int rr = lane >> 3, pcnk = lane & 7;
        // The DMA itself goes through inline asm (recipe of cdna_hip_programming.md section 5.7: M0 = wave-uniform LDS byte
        // address, saved and restored inside the statement): hipcc tracks the builtin form as a pending LDS write and puts
        // s_waitcnt vmcnt(0) in front of every later ds_read_b64_tr_b16 of the V image, which would drain the second phase at
        // the first P.V product. The waits that order these DMAs against the LDS reads are the two explicit ones below.
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
        auto glds16 = [&](const bf16_t* gsrc, unsigned lds_dst) {
            if constexpr (MERV_ATTN_ABL == 2) return;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
        };
        auto dma_k = [&](int pc) {
            const int row = pc * 8 + rr;
            const int grow = row < L ? row : L - 1;
            glds16(kbase + (size_t)grow * ld + ((pcnk ^ kswz(row)) * 8), lds0 + pc * 1024);
        };
        auto dma_v = [&](int pc) {
            const int row = pc * 8 + rr;
            const int grow = row < L ? row : L - 1;
            glds16(vbase + (size_t)grow * ld + ((pcnk ^ (((row >> 1) & 1) << 2)) * 8), lds0 + K_BYTES + pc * 1024);
        };
        static_assert(NW == 4 && RES_KROWS == 264 && RES_VROWS == 272, "piece schedule below is written for these sizes");
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {  // key tiles 0, 1: pieces 8 tq .. 8 tq + 7, two K and two V pieces per wave
            dma_k(8 * tq + wv); dma_k(8 * tq + 4 + wv);
            dma_v(8 * tq + wv); dma_v(8 * tq + 4 + wv);
        }
        // the BUILTIN form of the wait: hipcc's waitcnt pass sees it and retires the Q fragment loads in its own bookkeeping; after
        // an asm wait it would still count them and stall their first uses on the second-phase DMAs it cannot see
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) alone (gfx9 encoding: expcnt = 7, lgkmcnt = 15 untouched)
        __syncthreads();
#pragma unroll
        for (int tq = 2; tq < 4; ++tq) {
            dma_k(8 * tq + wv); dma_k(8 * tq + 4 + wv);
            dma_v(8 * tq + wv); dma_v(8 * tq + 4 + wv);
        }
        if (wv == 0) dma_k(32);          // K rows 256 .. 263
        if (wv < 2) dma_v(32 + wv);      // V rows 256 .. 271
    } else {
        load_tile(0);
    }
    MERV_STAMP(1);
    // Scores are wanted in log2 units, s' = (q . k) scale log2(e), so that the MFMA chain -- seeded with -m -- leaves the argument
    // of the exponential itself (round 4). The factor goes into Q once per block: 1 when the producer folded it into the q rows of
    // the qkv weight (AttnArgs::q_prescaled: no extra rounding anywhere), otherwise q is scaled and re-rounded to bf16 here.
    const float qs = p.q_prescaled ? 1.0f : sc;
    auto scale_q = [&](u32x4 w) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack2bf(bflo(w[j]) * qs, bfhi(w[j]) * qs);
        return __builtin_bit_cast(bf16x8, w);
    };
    bf16x8 qf[QPW][4], qx[4];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi)
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qi][s] = scale_q(qraw[qi][s]);
    if constexpr (XQ) {
#pragma unroll
        for (int s = 0; s < 4; ++s) qx[s] = scale_q(qxraw[s]);
    }
    const float psum_limit = fast_exp2(p.rescale_thr);  // a lane's 32 exponentials may sum to this before its reference moves
    // One key tile. TAILK (compile time): 0 = decide at run time whether the tile is the ragged last one; 1 = a full tile;
    // 2 = the resident kernel's last tile (sequences of 257 .. 264 tokens: 1 .. 8 keys, see tile_softmax).
    auto tile_body = [&](const int t, auto tailk_tag) {
        constexpr int TAILK = decltype(tailk_tag)::value;
        const int kv0 = t * 64;
        if constexpr (!RES) {
            __syncthreads();  // previous tile's LDS reads are done
            MERV_STAMP(2 + 4 * t);
            write_tile();
            MERV_STAMP(3 + 4 * t);
            __syncthreads();
            MERV_STAMP(4 + 4 * t);
            if (t + 1 < ntiles) load_tile(kv0 + 64);
        }
        if constexpr (RES) {
            if (t == 2) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // key tiles 2 .. 4 have landed (every wave's share)
        }
        const char* k_t = RES ? k_lds + kv0 * KROW : k_lds;  // this key tile's rows
        const char* v_t = RES ? v_lds + kv0 * 128 : v_lds;
        const bool tail = TAILK == 0 ? kv0 + 64 > L : TAILK == 2;
        // 257 = 4 * 64 + 1 and 3137 = 49 * 64 + 1: in the last tile of those sequences keys 32..63 are all padding;
        // their score MFMAs, exponentials and P.V MFMAs are skipped (wave-uniform)
        const bool both_halves = TAILK == 0 ? kv0 + 32 < L : TAILK == 1;

        // ---- S^T = K Q^T (keys on rows, queries on lanes). With <= 2 query tiles per wave all of them are issued
        //      first, so the MFMAs of tile qi+1 run in the matrix pipe under the softmax VALU work of tile qi; with 3
        //      tiles the score registers would not fit and each tile is multiplied right before its softmax ----
        constexpr bool S_FIRST = QPW <= 2;
        f32x16 sacc[S_FIRST ? QPW : 1][2];
        // seed: this lane's (= this query's) value for all 32 of its score registers: 0, or -m so that the chain ends on s' - m
        auto scores = [&](const bf16x8(&qfr)[4], f32x16(&sa)[2], const float seed) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both_halves) continue;
                const int key = kb * 32 + r;
                const char* krow = k_t + key * KROW;
                const int sw = kswz(key);
#pragma unroll
                for (int i = 0; i < 16; ++i) sa[kb][i] = seed;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 kf = *(const bf16x8*)(krow + (((2 * s + h) ^ sw) * 16));
                    sa[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfr[s], sa[kb], 0, 0, 0);
                }
            }
        };
        // One lane's share of a score tile (scores s' in log2 units, seed 0) -> exponentials against the exact running maximum in
        // place, returns their sum; m_new = max(m_old, tile max).
        // NE = 16: elements of both 32-key halves (the second only when it holds real keys), masked on the tail tile.
        // NE = 4 (the resident kernel's tail tile: sequences of 257 .. 264 tokens leave 1 .. 8 keys there, i.e. elements 0 .. 3 of
        // the first half in both half-waves): 4 masks / exponentials instead of 32 compares and 16 exponentials; elements 4 .. 7
        // are zeroed for the one P.V step that still runs (keys 0 .. 15 of the tile).
        auto mask_tail = [&](auto ne_tag, f32x16(&sa)[2]) {
            constexpr int NE = decltype(ne_tag)::value;
            constexpr int NKB = NE == 16 ? 2 : 1;
            if (tail) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int i = 0; i < NE; ++i) {
                        const int key = kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (key >= L) sa[kb][i] = -INFINITY;  // (covers the skipped half too: its registers are stale)
                    }
            }
        };
        auto tile_softmax = [&](auto ne_tag, f32x16(&sa)[2], float m_old, float& m_new) -> float {
            constexpr int NE = decltype(ne_tag)::value;
            constexpr int NKB = NE == 16 ? 2 : 1;
            float mx = -INFINITY;
            mask_tail(ne_tag, sa);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb == 1 && !both_halves) continue;
#pragma unroll
                for (int i = 0; i < NE; ++i) mx = fmaxf(mx, sa[kb][i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            m_new = fmaxf(mx, m_old);
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb == 1 && !both_halves) continue;
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const float e = fast_exp2(sa[kb][i] - m_new);
                    sa[kb][i] = e;
                    psum += e;
                }
            }
            if constexpr (NE < 8) {
#pragma unroll
                for (int i = NE; i < 8; ++i) sa[0][i] = 0.f;
            }
            return psum;
        };
        // The common form (deferred max, cdna_hip_programming.md T13, without a maximum): the chain was seeded with -m, the
        // registers hold s' - m, and the tile costs one v_exp and one v_add per score -- no v_max, no v_fma. The reference m only
        // has to keep the exponentials in range: a lane's 32 values are >= 0, so "their sum <= 2^thr" bounds every one of them
        // (fp32 sums and bf16 P keep their relative precision up to there); a larger or non-finite sum sends the WAVE back through
        // the exact form above for this tile (tile_body), which also moves m.
        auto tile_exp = [&](auto ne_tag, f32x16(&sa)[2]) -> float {
            constexpr int NE = decltype(ne_tag)::value;
            constexpr int NKB = NE == 16 ? 2 : 1;
            mask_tail(ne_tag, sa);
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb == 1 && !both_halves) continue;
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const float e = fast_exp2(sa[kb][i]);
                    sa[kb][i] = e;
                    psum += e;
                }
            }
            if constexpr (NE < 8) {
#pragma unroll
                for (int i = NE; i < 8; ++i) sa[0][i] = 0.f;
            }
            return psum;
        };
        // the first tile has no reference yet (m = -inf): seed 0 and the exact form
        if constexpr (S_FIRST) {
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) scores(qf[qi], sacc[qi], t == 0 ? 0.f : -m_run[qi]);
        }
        using NE_t = std::integral_constant<int, TAILK == 2 ? 4 : 16>;
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            if (!XQ && q_base + qi * 32 >= L) continue;  // wave-uniform: this query tile is entirely padding (XQ: the block's tiles are whole)
            f32x16(&sa)[2] = sacc[S_FIRST ? qi : 0];
            if constexpr (!S_FIRST) scores(qf[qi], sa, t == 0 ? 0.f : -m_run[qi]);
            // ---- online softmax (this lane: one query, 32 of the tile's 64 keys) ----
            float psum = 0.f;
            bool exact = t == 0;
            if (!exact) {
                psum = tile_exp(NE_t{}, sa);
                exact = !__all(psum <= psum_limit);  // (a NaN or an infinite sum fails the comparison too)
                if (exact) scores(qf[qi], sa, 0.f);  // rare: this tile again from its raw scores
            }
            if (exact) {
                float m_new;
                psum = tile_softmax(NE_t{}, sa, m_run[qi], m_new);
                if (t > 0 && !__all(m_new == m_run[qi])) {  // the running max moved for some query of this wave: rescale
                    const float alpha = fast_exp2(m_run[qi] - m_new);
                    l_run[qi] *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { oacc[qi][0][i] *= alpha; oacc[qi][1][i] *= alpha; }
                }
                m_run[qi] = m_new;  // (first tile: O and l are still zero, nothing to rescale)
            }
            l_run[qi] += psum;

            // ---- O^T += V^T P^T ----
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both_halves) continue;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int key_base = kb * 32 + 16 * s2;
                    if constexpr (RES && TAILK != 1) { if (kv0 + key_base >= L) continue; }  // all 16 keys are padding (P = 0) and their V rows are not resident (a full tile has none: no test, one basic block)
                    const bf16x8 pf = pack8(sa[kb], 8 * s2);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const bf16x8 vf = load_vt_frag<VTR, RES>(v_t, key_base, db, lane, VROWB);
                        oacc[qi][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[qi][db], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (XQ) {
            if (t % NW == wave) {  // wave-uniform: this wave multiplies the extra rows against key tile t, once, start to finish
                f32x16 sx[2];
                scores(qx, sx, 0.f);
                float mx;  // the tile's own maximum (finite: key kv0 of every tile is a real key); partials are merged after the loop
                const float psum = tile_softmax(std::integral_constant<int, TAILK == 2 ? 4 : 16>{}, sx, -INFINITY, mx);
                f32x16 ox[2];
#pragma unroll
                for (int i = 0; i < 16; ++i) { ox[0][i] = 0.f; ox[1][i] = 0.f; }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (kb == 1 && !both_halves) continue;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        if constexpr (RES && TAILK != 1) { if (kv0 + kb * 32 + 16 * s2 >= L) continue; }
                        const bf16x8 pf = pack8(sx[kb], 8 * s2);
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const bf16x8 vf = load_vt_frag<VTR, RES>(v_t, kb * 32 + 16 * s2, db, lane, VROWB);
                            ox[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ox[db], 0, 0, 0);
                        }
                    }
                }
                if (r < L - XQ0) {  // this lane's query is a real row: its 32 d-values, the tile's max and its half of the sum
                    float* dst = xq_lds + (t * XQ_ROWS + r) * (HD + 4);
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int i = 0; i < 16; ++i) dst[db * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] = ox[db][i];
                    dst[HD + 1 + h] = psum;
                    if (h == 0) dst[HD] = mx;
                }
            }
        }
        MERV_STAMP(5 + 4 * t);
    };
    if constexpr (MERV_ATTN_ABL == 1) {
    } else if constexpr (RES) {  // L in (256, 264]: four full tiles, then the 1 .. 8 keys of the fifth
        for (int t = 0; t < ntiles - 1; ++t) tile_body(t, std::integral_constant<int, 1>{});
        tile_body(ntiles - 1, std::integral_constant<int, 2>{});
    } else {
        // every tile but the last is a full one, known at compile time: its body has no tail / half-tile branches, so hipcc keeps the two
        // query tiles' score chains interleaved and reads each K fragment once for both (with the run-time form every 32-key half sat
        // behind a branch: fragments re-read per query tile, each read waited for right in front of its MFMA)
        for (int t = 0; t < ntiles - 1; ++t) tile_body(t, std::integral_constant<int, 1>{});
        tile_body(ntiles - 1, std::integral_constant<int, 0>{});
    }
    MERV_STAMP(30);

    // ---- output: transpose each 32 x 64 tile through LDS so the global stores are 16 B per lane, 128 B per row ----
    __syncthreads();  // every wave is done with the K/V tiles (and, XQ, has left its partials of the extra rows)
    if constexpr (XQ) {
        // merge the key-tile partials of extra row `row`: lane = output column
        for (int row = wave; row < L - XQ0; row += NW) {
            float m = -INFINITY;
            for (int t = 0; t < ntiles; ++t) m = fmaxf(m, xq_lds[(t * XQ_ROWS + row) * (HD + 4) + HD]);
            float o = 0.f, l = 0.f;
            for (int t = 0; t < ntiles; ++t) {
                const float* src = xq_lds + (t * XQ_ROWS + row) * (HD + 4);
                const float w = fast_exp2(src[HD] - m);
                o = fmaf(w, src[lane], o);
                l = fmaf(w, src[HD + 1] + src[HD + 2], l);
            }
            const bf16_t ob = f2bf(o / l);
            if constexpr (MXQ) {  // MXFP8 output: the row's 64 columns are two 32-column blocks, one per half-wave (lane = column)
                const float f = bf2f(ob);
                float amax = fabsf(f);
#pragma unroll
                for (int sh = 1; sh < 32; sh <<= 1) amax = fmaxf(amax, __shfl_xor(amax, sh, 64));
                const int e = mx_shared_exponent(amax);
                const float inv = __uint_as_float((uint32_t)(127 - e) << 23);
                const int grow = seq * L + XQ0 + row, col = head * HD + lane;
                p.mx_q[(size_t)grow * D + col] = (uint8_t)(mx_pack4(f, 0.f, 0.f, 0.f, inv) & 0xffu);
                if ((lane & 31) == 0) p.mx_scales[mx_scale_offset(grow, col >> 5, p.mx_groups)] = (uint8_t)(e + 127);
            } else {
                p.out[((size_t)seq * L + XQ0 + row) * D + head * HD + lane] = ob;
            }
        }
    }
    char* stg = smem + wave * (32 * 128);
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int q0 = q_base + qi * 32;
        if (q0 >= L) continue;
        const float l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32, 64);
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x2 o;
                o[0] = pack2bf(oacc[qi][db][4 * i + 0] * inv, oacc[qi][db][4 * i + 1] * inv);
                o[1] = pack2bf(oacc[qi][db][4 * i + 2] * inv, oacc[qi][db][4 * i + 3] * inv);
                *(u32x2*)(stg + r * 128 + (((db * 4 + i) ^ (r & 7)) * 16) + 8 * h) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private staging: in-wave ordering suffices
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = (lane >> 3) + 8 * it, c = lane & 7;
            const u32x4 v = *(const u32x4*)(stg + rr * 128 + ((c ^ (rr & 7)) * 16));
            if constexpr (MXQ) {  // lanes c = 4b .. 4b+3 hold the 32-column block b of this head's row
                float f[8];
#pragma unroll
                for (int w = 0; w < 4; ++w) { f[2 * w] = bflo(v[w]); f[2 * w + 1] = bfhi(v[w]); }
                int sb;
                const u32x2 q8 = mx_quantize8(f, sb);
                if (q0 + rr < L) {
                    const int row = seq * L + q0 + rr, col = head * HD + c * 8;
                    *(u32x2*)(p.mx_q + (size_t)row * D + col) = q8;
                    if ((c & 3) == 0) p.mx_scales[mx_scale_offset(row, col >> 5, p.mx_groups)] = (uint8_t)sb;
                }
            } else if ((XQ || q0 + rr < L) && (MERV_ATTN_ABL != 3 || p.L < 0)) {  // (XQ: the block's eight tiles are whole, the rows past them go the other way)
                *(u32x4*)(p.out + ((size_t)seq * L + q0 + rr) * D + head * HD + c * 8) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads returned before the next tile overwrites
    }
    MERV_STAMP(31);
}

// ---------------------------------------------------------------------------------------------------------
// Causal self attention of the LLM prompt prefill (row f-3): head_dim = 128, one sequence, GQA. The same transposed-score scheme
// as attn_kernel -- S^T = K Q^T per 64-key tile (8 k-steps), one query per lane, P already in B-operand order for O^T = V^T P^T
// (4 d-blocks), deferred running maximum with the chain seeded by -m -- re-parameterised: K / V rows are 256 B (the KV cache's
// rows), a block is 4 waves x 32 queries of one head, and the key-tile loop of a block stops at its last query's tile; inside the
// diagonal tiles keys past the lane's query are masked. Blocks are dispatched longest-first (the last query block has the most
// key tiles). Replaces F.scaled_dot_product_attention(..., is_causal=True) inside LlamaAttention.forward for the prompt
// (merv/models/vidlms/merv.py:723-734 -> LlamaForCausalLM.forward): 145 us per layer there at 1049 tokens x 32 heads.
// ---------------------------------------------------------------------------------------------------------
constexpr int PA_HD = 128, PA_KROW = 256, PA_VROW = 320;  // V rows padded to 320 B: four rows' 64-byte column groups tile the 256-B bank row
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void prefill_attn_kernel(PrefillAttnArgs p) {
    constexpr int NT = NW * 64;
    constexpr int STG = 1024 / NT;  // 16-byte chunks of a 64 x 128 bf16 tile per thread
    constexpr int K_BYTES = 64 * PA_KROW, V_BYTES = 64 * PA_VROW, OUT_BYTES = NW * 32 * 256;
    constexpr int LDS_BYTES = (K_BYTES + V_BYTES) > OUT_BYTES ? (K_BYTES + V_BYTES) : OUT_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    char* k_lds = smem;
    char* v_lds = smem + K_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, kvh = head / (p.H / p.Hkv);
    const int qb = (int)gridDim.x - 1 - (int)blockIdx.x;  // longest blocks first
    const int S = p.S;
    const int q0 = qb * (NW * 32), qw = q0 + wave * 32;  // first query of the block / of this wave
    const bf16_t* kbase = p.k + (size_t)kvh * p.kv_head_stride;
    const bf16_t* vbase = p.v + (size_t)kvh * p.kv_head_stride;

    // Q^T B-operand fragments: element j of step s = Q[q][16 s + 8 h + j], scaled to log2 units (scale * log2 e, re-rounded to bf16)
    const float sc = p.scale * LOG2E;
    bf16x8 qf[8];
    {
        const int q_row = qw + r < S ? qw + r : S - 1;
        const bf16_t* qp = p.q + (size_t)q_row * p.ldq + head * PA_HD + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            u32x4 w = *(const u32x4*)(qp + 16 * s);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = pack2bf(bflo(w[j]) * sc, bfhi(w[j]) * sc);
            qf[s] = __builtin_bit_cast(bf16x8, w);
        }
    }
    f32x16 oacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float psum_limit = fast_exp2(p.rescale_thr);

    u32x4 kreg[STG], vreg[STG];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < STG; ++i) {
            const int idx = tid + NT * i;
            int key = kv0 + (idx >> 4);
            key = key < S ? key : S - 1;
            kreg[i] = *(const u32x4*)(kbase + (size_t)key * p.ldk + (idx & 15) * 8);
            vreg[i] = *(const u32x4*)(vbase + (size_t)key * p.ldk + (idx & 15) * 8);
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < STG; ++i) {
            const int idx = tid + NT * i, key = idx >> 4, c = idx & 15;
            *(u32x4*)(k_lds + key * PA_KROW + ((c ^ (key & 15)) * 16)) = kreg[i];  // 16 consecutive rows -> 16 distinct 16-B slots
            *(u32x4*)(v_lds + key * PA_VROW + c * 16) = vreg[i];
        }
    };
    // key tiles 0 .. the tile of the block's last (real) query
    const int q_last = (q0 + NW * 32 - 1 < S ? q0 + NW * 32 - 1 : S - 1);
    const int ntiles = q_last / 64 + 1;
    load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * 64;
        __syncthreads();  // previous tile's LDS reads are done
        write_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile(kv0 + 64);
        if (kv0 > qw + 31 || qw >= S) continue;  // wave-uniform: every key of this tile is past the wave's queries (or the wave is padding)
        const bool both_halves = kv0 + 32 <= qw + 31;
        const bool diag = kv0 + 63 > qw;  // some (query, key) pairs of the tile are masked
        f32x16 sa[2];
        auto scores = [&](const float seed) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both_halves) continue;
                const int key = kb * 32 + r;
                const char* krow = k_lds + key * PA_KROW;
                const int sw = key & 15;
#pragma unroll
                for (int i = 0; i < 16; ++i) sa[kb][i] = seed;
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const bf16x8 kf = *(const bf16x8*)(krow + (((2 * s8 + h) ^ sw) * 16));
                    sa[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s8], sa[kb], 0, 0, 0);
                }
            }
        };
        auto mask = [&]() {
            if (diag) {
                const int qry = qw + r;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (key > qry) sa[kb][i] = -INFINITY;  // (covers a skipped second half too: its registers are stale)
                    }
            }
        };
        scores(t == 0 ? 0.f : -m_run);
        float psum = 0.f;
        bool exact = t == 0;
        if (!exact) {  // common form: the chain left s' - m, one v_exp + one v_add per score (see attn_kernel)
            mask();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both_halves) continue;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = fast_exp2(sa[kb][i]);
                    sa[kb][i] = e;
                    psum += e;
                }
            }
            exact = !__all(psum <= psum_limit);
            if (exact) scores(0.f);
        }
        if (exact) {
            mask();
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both_halves) continue;
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sa[kb][i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(mx, m_run);  // finite from tile 0 on: key 0 is visible to every query
            psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both_halves) continue;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = fast_exp2(sa[kb][i] - m_new);
                    sa[kb][i] = e;
                    psum += e;
                }
            }
            if (t > 0 && !__all(m_new == m_run)) {
                const float alpha = fast_exp2(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
            }
            m_run = m_new;
        }
        l_run += psum;
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && !both_halves) continue;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int key_base = kb * 32 + 16 * s2;
                if (kv0 + key_base > qw + 31) continue;  // wave-uniform: these 16 keys are past every query of the wave (P = 0)
                const bf16x8 pf = pack8(sa[kb], 8 * s2);
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const bf16x8 vf = load_vt_frag<true, false>(v_lds, key_base, db, lane, PA_VROW);
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[db], 0, 0, 0);
                }
            }
        }
    }
    // ---- output: transpose the wave's 32 x 128 tile through LDS: 16 B per lane, 256 B per row ----
    __syncthreads();
    if (qw >= S) return;
    char* stg = smem + wave * (32 * 256);
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 o;
            o[0] = pack2bf(oacc[db][4 * i + 0] * inv, oacc[db][4 * i + 1] * inv);
            o[1] = pack2bf(oacc[db][4 * i + 2] * inv, oacc[db][4 * i + 3] * inv);
            // d = 32 db + 8 i + 4 h + 0..3 of query r: 16-byte chunk 4 db + i, half h; chunks XOR-swizzled by the row
            *(u32x2*)(stg + r * 256 + (((db * 4 + i) ^ (r & 15)) * 16) + 8 * h) = o;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private staging: in-wave ordering suffices
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int rr = (lane >> 4) + 4 * it, c = lane & 15;
        const u32x4 v = *(const u32x4*)(stg + rr * 256 + ((c ^ (rr & 15)) * 16));
        if (qw + rr < S) *(u32x4*)(p.out + (size_t)(qw + rr) * p.ldo + head * PA_HD + c * 8) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// LanguageBind temporal attention (modeling_video.py:133-155): for each (clip, token, head) attend over the
// clip's t = 8 frames. Four such 8x8 problems are packed block-diagonally into one 32x32 MFMA tile per wave
// (off-diagonal blocks masked to -inf), so the whole sub-block costs 8 MFMAs per 4 problems.
// ---------------------------------------------------------------------------------------------------------
constexpr int TV_ROW_TR = 192;  // row-major V image per wave: 32 rows x 192 B
constexpr int TVT_ROW = 72;     // V^T image per wave: 64 d-rows x (32 keys * 2 B + 8 B pad)

// NH: heads per block = consecutive waves that take the SAME four problems (32 rows): their loads are NH x 128 contiguous bytes per
// row and tensor instead of 128 (DRAM pages, round 4: 111 against 122 us per layer at 16 videos with NH = 8).
template <bool VTR, int NH>
__global__ __launch_bounds__(NH > 4 ? NH * 64 : 256) void temporal_attn_kernel(TemporalAttnArgs p) {
    constexpr int NWAVE = NH > 4 ? NH : 4;
    constexpr int WAVE_LDS = VTR ? 32 * TV_ROW_TR : 64 * TVT_ROW;
    constexpr int VROWB = VTR ? TV_ROW_TR : TVT_ROW;
    __shared__ __attribute__((aligned(16))) char smem[NWAVE * WAVE_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y * NH + wave % NH;
    const int D = p.D, ld = 3 * D;
    const int NP = p.nclips * p.ntok;           // number of (clip, token) problems
    const int pid0 = (blockIdx.x * (NWAVE / NH) + wave / NH) * 4;  // first of this wave's 4 problems
    char* v_lds = smem + wave * WAVE_LDS;

    auto row_of = [&](int idx) -> size_t {  // idx in [0,32): problem idx>>3, frame idx&7
        int pid = pid0 + (idx >> 3);
        pid = pid < NP ? pid : NP - 1;
        const int clip = pid / p.ntok, tok = pid - clip * p.ntok;
        return (size_t)(clip * 8 + (idx & 7)) * p.ntok + tok;
    };
    const size_t my_row = row_of(r);

    bf16x8 qf[4], kf[4];
    if constexpr (VTR) {
        // Every global access is 8 lanes x 16 B = one 128-byte head segment of a row (round 4; the fragment-shaped loads were 32-byte
        // pieces of 32 different rows per instruction). q, k and v rows arrive in registers in that shape; q and k then pass through
        // the wave's LDS region one after the other (chunks XOR-swizzled by the row, as the K tiles of attn_kernel) to be read back
        // as MFMA fragments, and the region finally holds the V image. Wave-private: in-wave ordering (lgkmcnt) is enough.
        u32x4 raw[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            const bf16_t* src = p.qkv + row_of(idx >> 3) * ld + head * HD + (idx & 7) * 8;
            raw[0][i] = *(const u32x4*)src;
            raw[1][i] = *(const u32x4*)(src + D);
            raw[2][i] = *(const u32x4*)(src + 2 * D);
        }
        auto through_lds = [&](const u32x4(&rw)[4], bf16x8(&frag)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i, row = idx >> 3, c = idx & 7;
                *(u32x4*)(v_lds + row * 128 + ((c ^ kswz(row)) * 16)) = rw[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < 4; ++s) frag[s] = *(const bf16x8*)(v_lds + r * 128 + (((2 * s + h) ^ kswz(r)) * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read back before the region is overwritten
        };
        through_lds(raw[0], qf);
        through_lds(raw[1], kf);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            *(u32x4*)(v_lds + (idx >> 3) * TV_ROW_TR + (idx & 7) * 16) = raw[2][i];
        }
    } else {
        const bf16_t* qrow = p.qkv + my_row * ld + head * HD;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *(const bf16x8*)(qrow + 16 * s + 8 * h);
            kf[s] = *(const bf16x8*)(qrow + D + 16 * s + 8 * h);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kp = (lane >> 3) + 8 * i, c = lane & 7;
            const u32x4 a = *(const u32x4*)(p.qkv + row_of(2 * kp) * ld + 2 * D + head * HD + c * 8);
            const u32x4 b = *(const u32x4*)(p.qkv + row_of(2 * kp + 1) * ld + 2 * D + head * HD + c * 8);
            write_vt_pair(v_lds, TVT_ROW, kp, c, a, b);
        }
    }

    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], sacc, 0, 0, 0);

    // this lane's query belongs to problem r>>3; key of register i is (i&3) + 8*(i>>2) + 4*h -> problem i>>2
    const float sc = p.scale * LOG2E;
    const int myp = r >> 3;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = ((i >> 2) == myp) ? sacc[i] * sc : -INFINITY;
        sacc[i] = v;
        mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float e = fast_exp2(sacc[i] - mx);
        sacc[i] = e;
        psum += e;
    }
    psum += __shfl_xor(psum, 32, 64);
    const float inv = 1.0f / psum;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] *= inv;

    __syncthreads();  // V image written (each wave only reads its own region)

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack8(sacc, 8 * s2);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const bf16x8 vf = load_vt_frag<VTR>(v_lds, 16 * s2, db, lane, VROWB);
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[db], 0, 0, 0);
        }
    }
    if (p.mx_q) {
        // MXFP8 output: the 32-column block `db` of a row is this lane's 16 values (4 per i) and the 16 of lane ^ 32
        const bool ok = pid0 + myp < NP;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float f[16];
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t w0 = pack2bf(oacc[db][4 * i + 0], oacc[db][4 * i + 1]);  // quantise the bf16-rounded values
                const uint32_t w1 = pack2bf(oacc[db][4 * i + 2], oacc[db][4 * i + 3]);
                f[4 * i] = bflo(w0); f[4 * i + 1] = bfhi(w0); f[4 * i + 2] = bflo(w1); f[4 * i + 3] = bfhi(w1);
#pragma unroll
                for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(f[4 * i + j]));
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            const int e = mx_shared_exponent(amax);
            const float inv = __uint_as_float((uint32_t)(127 - e) << 23);
            if (ok) {
                const int col0 = head * HD + db * 32;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *(uint32_t*)(p.mx_q + (size_t)my_row * D + col0 + 8 * i + 4 * h) =
                        mx_pack4(f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3], inv);
                if (h == 0) p.mx_scales[mx_scale_offset((int)my_row, col0 >> 5, p.mx_groups)] = (uint8_t)(e + 127);
            }
        }
    } else if constexpr (VTR) {
        // transpose the wave's 32 x 64 result through its LDS region (the V image is consumed): stores of 16 B per lane, 128 B per row
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x2 o;
                o[0] = pack2bf(oacc[db][4 * i + 0], oacc[db][4 * i + 1]);
                o[1] = pack2bf(oacc[db][4 * i + 2], oacc[db][4 * i + 3]);
                *(u32x2*)(v_lds + r * 128 + (((db * 4 + i) ^ (r & 7)) * 16) + 8 * h) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = (lane >> 3) + 8 * it, c = lane & 7;
            const u32x4 v = *(const u32x4*)(v_lds + rr * 128 + ((c ^ (rr & 7)) * 16));
            if (pid0 + (rr >> 3) < NP) *(u32x4*)(p.out + row_of(rr) * D + head * HD + c * 8) = v;
        }
    } else if (pid0 + myp < NP) {
        bf16_t* orow = p.out + my_row * D + head * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x2 o;
                o[0] = pack2bf(oacc[db][4 * i + 0], oacc[db][4 * i + 1]);
                o[1] = pack2bf(oacc[db][4 * i + 2], oacc[db][4 * i + 3]);
                *(u32x2*)(orow + db * 32 + 8 * i + 4 * h) = o;
            }
    }
}

}  // namespace

static float initial_rescale_thr() {
    const char* e = merv_tuning_env("MERV_ATTN_RESCALE_THR");
    const float v = e ? (float)atof(e) : 8.0f;
    return (v >= 0.f && v <= 64.f) ? v : 8.0f;
}
static std::atomic<float> g_attn_rescale_thr{initial_rescale_thr()};
void set_attn_rescale_thr(float thr) {  // (product build: a no-op -- see merv_tuning_env, common.h)
    if constexpr (MERV_HOOKS) g_attn_rescale_thr.store((thr >= 0.f && thr <= 64.f) ? thr : 8.0f, std::memory_order_relaxed);
}

// MERV_ATTN_VTR=0 in the environment selects the transposing-store V path (diagnostic switch, re-read per launch).
static bool use_vtr() {
    const char* e = merv_tuning_env("MERV_ATTN_VTR");
    return !(e && e[0] == '0');
}

template <int NW, int QPW, bool XQ = false, bool RES = false>
static hipError_t launch_attn_cfg(const AttnArgs& a, hipStream_t s) {
    const int rows = NW * QPW * 32;
    dim3 grid(XQ ? 1 : (a.L + rows - 1) / rows, a.heads, a.nseq);
    ProfScope pk(RES ? PROF_K_ATTN_RES : PROF_K_ATTN_STREAM, s, 4.0 * a.nseq * (double)a.L * a.L * a.D, 2.0 * 4.0 * a.nseq * (double)a.L * a.D);
    if (a.mx_q) {  // MXFP8 output (opt-in mode): its own instantiations
        if (RES || use_vtr()) hipLaunchKernelGGL((attn_kernel<true, NW, QPW, XQ, RES, true>), grid, dim3(NW * 64), 0, s, a);
        else if constexpr (!RES) hipLaunchKernelGGL((attn_kernel<false, NW, QPW, XQ, false, true>), grid, dim3(NW * 64), 0, s, a);
    } else if (RES || use_vtr())
        hipLaunchKernelGGL((attn_kernel<true, NW, QPW, XQ, RES>), grid, dim3(NW * 64), 0, s, a);
    else if constexpr (!RES)
        hipLaunchKernelGGL((attn_kernel<false, NW, QPW, XQ, false>), grid, dim3(NW * 64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_attention(const AttnArgs& a_in, hipStream_t s) {
    if (a_in.nseq <= 0 || a_in.L <= 0) return hipSuccess;
    if (a_in.D != a_in.heads * HD) return hipErrorInvalidValue;
    AttnArgs a = a_in;
    // deferred-max threshold in binary orders of magnitude (a lane's exponentials may sum to 2^thr before its reference moves);
    // merv_debug_set_attn_rescale_thr (tests, probes) overrides the default, MERV_ATTN_RESCALE_THR sets it once per process
    a.rescale_thr = g_attn_rescale_thr.load(std::memory_order_relaxed);
    ProfScope ps(PROF_ATTN, s, 4.0 * a.nseq * (double)a.L * a.L * a.D, 2.0 * 4.0 * a.nseq * (double)a.L * a.D);
    // block shape. Short sequences: least padded query tiles -- 257- / 261-token sequences are exactly 9 tiles = 3 waves
    // x 3, streamed once per (sequence, head). Long sequences (ViViT, 3137 tokens = 99 tiles): 4 waves x 2 even at 5 %
    // more padded tiles, because with two tiles per wave both score products are issued before the first softmax, so
    // the matrix pipe works under the softmax VALU stream (measured at B=8: 354 us vs 440-463 us per layer with 3 x 3).
    const int t32 = (a.L + 31) / 32;
    static const char* force = merv_tuning_env("MERV_ATTN_CFG");  // tuning hook: "33" or "42"
    if (force && force[0] == '3') return launch_attn_cfg<3, 3>(a, s);
    if (force && force[0] == '4') return launch_attn_cfg<4, 2>(a, s);
    if (t32 <= 4) return launch_attn_cfg<4, 1>(a, s);
    // 8 full tiles + 1..8 rows (257 / 261 tokens): 4 x 2 block, the extra rows split over the waves by key tile
    if (a.L > 256 && a.L <= 256 + XQ_ROWS) {  // (MXFP8 output too since round 6: the extra rows' merge quantises its row per half-wave)
        if (force && force[0] == 'x') return launch_attn_cfg<4, 2, true>(a, s);  // streamed K / V tiles (A/B of the resident form)
        if (use_vtr()) return launch_attn_cfg<4, 2, true, true>(a, s);           // all K / V rows resident in LDS
        return launch_attn_cfg<4, 2, true>(a, s);
    }
    if (a.L >= 1024) return launch_attn_cfg<4, 2>(a, s);
    const int pad9 = (t32 + 8) / 9 * 9 - t32, pad8 = (t32 + 7) / 8 * 8 - t32;
    if (pad9 < pad8) return launch_attn_cfg<3, 3>(a, s);
    return launch_attn_cfg<4, 2>(a, s);
}

hipError_t launch_prefill_attention(const PrefillAttnArgs& a_in, hipStream_t s) {
    if (a_in.S <= 0) return hipSuccess;
    if (a_in.H <= 0 || a_in.Hkv <= 0 || a_in.H % a_in.Hkv != 0) return hipErrorInvalidValue;
    if ((a_in.ldq | a_in.ldk | a_in.ldo) % 8 != 0 || a_in.kv_head_stride % 8 != 0) return hipErrorInvalidValue;
    PrefillAttnArgs a = a_in;
    a.rescale_thr = g_attn_rescale_thr.load(std::memory_order_relaxed);
    constexpr int NW = 4;
    const int nqb = (a.S + NW * 32 - 1) / (NW * 32);
    hipLaunchKernelGGL(prefill_attn_kernel<NW>, dim3(nqb, a.H), dim3(NW * 64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_temporal_attention(const TemporalAttnArgs& a, hipStream_t s) {
    if (a.nclips <= 0) return hipSuccess;
    if (a.t != 8 || a.D != a.heads * HD) return hipErrorInvalidValue;
    const int NP = a.nclips * a.ntok;
    ProfScope ps(PROF_TATTN, s, 4.0 * NP * 64.0 * a.D, 2.0 * 4.0 * NP * 8.0 * a.D);
    static const char* force = merv_tuning_env("MERV_TATTN_NH");  // tuning hook: heads per block 1 / 4 / 8
    int nh = a.heads % 8 == 0 ? 8 : (a.heads % 4 == 0 ? 4 : 1);  // 16 videos: 122 / 113.5 / 111 us per layer with 1 / 4 / 8 heads per block
    if (force && (force[0] == '1' || (force[0] == '4' && a.heads % 4 == 0) || (force[0] == '8' && a.heads % 8 == 0))) nh = force[0] - '0';
    auto go = [&](auto nh_tag) {
        constexpr int NH = decltype(nh_tag)::value;
        constexpr int NWAVE = NH > 4 ? NH : 4;
        const int ppb = (NWAVE / NH) * 4;  // problems per block
        dim3 grid((NP + ppb - 1) / ppb, a.heads / NH);
        if (use_vtr()) hipLaunchKernelGGL((temporal_attn_kernel<true, NH>), grid, dim3(NWAVE * 64), 0, s, a);
        else hipLaunchKernelGGL((temporal_attn_kernel<false, NH>), grid, dim3(NWAVE * 64), 0, s, a);
    };
    if (nh == 8) go(std::integral_constant<int, 8>{});
    else if (nh == 4) go(std::integral_constant<int, 4>{});
    else go(std::integral_constant<int, 1>{});
    return hipGetLastError();
}

}  // namespace merv
