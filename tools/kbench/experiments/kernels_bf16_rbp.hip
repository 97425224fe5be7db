// Fused ResBlock1 pair in bf16 as a PERSISTENT, SOFTWARE-PIPELINED kernel:   x' = c2(lrelu(c1(lrelu(x)))) + x
// (vietTTS/hifigan/model.py:45-50).  Third generation of the pair kernel (kernels_bf16_rbg.hip is the second).
//
// What round 1's per-workgroup timelines established (profiles/r01_j_kbench_findings.md): with two workgroups per CU the
// staging / epilogue phases of one workgroup share each SIMD's issue port with the other workgroup's MFMA stream and get
// about ONE VALU slot per MFMA; a tile's ~1400 VALU instructions per wave then take as long as its 1408 MFMAs, and the
// matrix pipe idles whenever both workgroups are in such a phase (~25 % of the time, MfmaUtil 0.65).  Inside ONE wave the
// same SIMD issues up to ~5 other instructions per MFMA for free (MI355X_MICROARCH.md, per-instruction constants).  So:
//
//   * ONE 4-wave workgroup per CU (one wave per SIMD, 512 registers), persistent: workgroup w walks a contiguous run of
//     tiles (neighbouring tiles of an utterance on the same CU back to back: the halo rows are L2 hits);
//   * tile i's two MFMA loops carry the neighbouring tiles' VALU work as FILLERS, a fixed quantum per block of 8 k-steps:
//         c1 loop of tile i:  epilogue 2 of tile i-1 (residual add, bf16 pack, 16-byte stores) — the c2 accumulators of
//                             tile i-1 stay live in a second accumulator set while c1 of tile i accumulates in the first;
//                             its last block issues the LDS-DMA of tile i+1's X rows (raw bf16, no registers);
//         c2 loop of tile i:  LeakyReLU of tile i+1's X rows IN PLACE in LDS (ds_read_b128 -> 24 VALU -> ds_write_b128, each
//                             lane on the 16 bytes it DMA'd); its last block requests tile i's residual rows;
//     only epilogue 1 (c1 accumulators -> LeakyReLU -> bf16 -> xt tile in LDS, ~320 VALU) stays exposed between two barriers;
//   * two LDS tile buffers (X(i) -> xt(i) in one, X(i+1) landing in the other): 2 x 80 KiB = the whole LDS at k = 11, rate 5;
//   * A operands (weights) as in the second generation: host-packed fragments straight from L2 into a register ring, ONE
//     continuous stream over both convolutions and across tiles; B operands from the LDS tile, XOR-swizzled.
//   * slow loads only where nothing waits behind them: VMEM returns in order, so a streaming HBM load issued in front of the
//     weight stream would stall the MFMA loop for its whole latency.  The DMA burst goes out in the last block of the c1 loop
//     (behind it: epilogue 1), the residual burst in the last block of the c2 loop (L2 / MALL hits: the rows were DMA'd one
//     tile earlier).
//
// The MRF read-modify-write pairs (acc_add: 2 of a stage's 9 pair launches) stay on the second-generation kernel: their
// second row stream would need another 64 registers in flight.
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "bf16_common.h"

#ifndef PEXP  // kernel-development switches (tools/kbench): bit 0 = no fillers, bit 1 = no DMA / residual bursts, bit 2 = B reads late in the step (first cut), bit 3 = no epilogue-2 stores, bit 4 = c2 accumulators not pinned to AGPRs,
              // bit 5 (round 6) = EXPLICIT issue order: every MFMA followed by its memory instruction and a 1-4 instruction slice of a filler unit, fenced by sched_barrier(0)
#define PEXP 0
#endif

namespace vtts {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(integral_constant<int, 0>{}) ... f(integral_constant<int, N - 1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int C_, int KS_, int DIL_, int N1_, int WM_, int WN_, int PA_, int QE_, int QL_>
struct PTile {
    static constexpr int C = C_, KS = KS_, DIL = DIL_, N1 = N1_, WM = WM_, WN = WN_, PA = PA_, QE = QE_, QL = QL_;
    static constexpr int THREADS = 64 * WM * WN, NWAVES = WM * WN;
    static constexpr int MR = C / WM / 32, NR = N1 / WN / 32;
    static constexpr int H2 = (KS - 1) / 2, H1 = H2 * DIL;
    static constexpr int NT2 = N1 - 2 * H2;            // outputs per tile
    static constexpr int SPR = C / 8, P = C * 2;       // 16-byte slots / bytes per tile row (X and xt alike)
    static constexpr int KSTEPS = C / 16;              // k-steps per tap
    static constexpr int NQ = KS * KSTEPS;             // k-steps per convolution
    static constexpr int MB = C / 32;
    static constexpr int RA = PA + 1;                  // A-fragment register ring (slots)
    static constexpr int UB = 8;                       // k-steps per block
    static constexpr int NBLK = NQ / UB;
    static constexpr int ROWSX = N1 + 2 * H1;          // X rows a tile reads
    static constexpr int ROWST = N1 + 2 * H2;          // xt rows incl. the tail only discarded columns read
    static constexpr int RPD = 1024 / P;               // tile rows per LDS-DMA instruction (1 KiB per wave-instruction)
    static constexpr int NDW_RAW = (ROWSX + RPD * NWAVES - 1) / (RPD * NWAVES);
    static constexpr int NDW = (NDW_RAW + QL - 1) / QL * QL;  // DMA instructions (= in-place LeakyReLU units) per wave and tile
    static constexpr int NBL = NDW / QL;               // c2-loop blocks that carry LeakyReLU units
    static constexpr int ROWS_BUF = NDW * NWAVES * RPD > ROWST ? NDW * NWAVES * RPD : ROWST;
    static constexpr int BUF_BYTES = ROWS_BUF * P;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static constexpr int NE2 = MR * 2 * NR;            // epilogue-2 units (16-byte chunks) per lane
    static constexpr int NBE = (NE2 + QE - 1) / QE;    // c1-loop blocks that carry epilogue-2 units
    static constexpr size_t CONV_BYTES = (size_t)KS * C * C * 2;
    static_assert(THREADS == 256, "one wave per SIMD");
    static_assert(C % (WM * 32) == 0 && N1 % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(KSTEPS == UB, "a block is one tap (C = 128)");
    static_assert((RA == 4 || RA == 8) && UB % RA == 0, "ring slot of a step is its position in the block");
    static_assert(NBE <= NBLK - 1 && NBL <= NBLK - 1, "the last block of each loop carries the burst");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(SPR == 16, "swizzle / DMA lane map written for 256-byte rows");
};

// tile -> (utterance, first output row); tiles past an utterance's end (ragged batches) are skipped
struct PTileRef {
    int b, t0, L;
};

// wave-uniform value -> SGPR
__device__ __forceinline__ int sgpr_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// keep the compiler from hoisting / tabulating what is derived from v (recompute it here, every time)
__device__ __forceinline__ void opaque(unsigned& v) { asm volatile("" : "+v"(v)); }

// LDS-DMA, 16 bytes per lane: lane's source = sbase (wave-uniform 64-bit) + voff; lands at LDS byte lds_dst + 16 * lane.
// Invisible to hipcc's wait counting (guide §5.7): the caller waits (vmcnt) and synchronises.
__device__ __forceinline__ void glds16_saddr(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    // s_nop 4: an SGPR operand may come straight from v_readfirstlane (VALU write -> VMEM read of an SGPR: 5 wait states, which
    // hipcc does not insert inside an asm statement, guide §5.7)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

template <class T>
__global__ __launch_bounds__(T::THREADS, 1) void resblock_pair_p_bf16_k(BConvArgs a, int tpu, int ntiles, int run) {
    constexpr int C = T::C, KS = T::KS, DIL = T::DIL, N1 = T::N1, WN = T::WN, PA = T::PA, RA = T::RA;
    constexpr int MR = T::MR, NR = T::NR, H2 = T::H2, H1 = T::H1, NT2 = T::NT2, SPR = T::SPR, P = T::P;
    constexpr int NQ = T::NQ, MB = T::MB, UB = T::UB, NBLK = T::NBLK, RPD = T::RPD, NDW = T::NDW;
    constexpr int NE2 = T::NE2, QE = T::QE, QL = T::QL, NBE = T::NBE, NBL = T::NBL, NWAVES = T::NWAVES;
    constexpr unsigned DROP = 0x80000000u;  // a buffer offset beyond every utterance: the access is dropped / reads 0

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sgpr_i(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int Lp = a.L;

    const int tile_lo = blockIdx.x * run, tile_hi = min(tile_lo + run, ntiles);
    auto tile_ref = [&](int tile) {
        PTileRef r;
        r.b = tile / tpu;
        r.t0 = (tile - r.b * tpu) * NT2;
        r.L = a.lens ? min(max(a.lens[r.b], 0) * a.len_mul, Lp) : Lp;
        r.b = sgpr_i(r.b);
        r.t0 = sgpr_i(r.t0);
        r.L = sgpr_i(r.L);
        return r;
    };
    auto next_valid = [&](int tile) {  // first tile >= `tile` of this workgroup's run that lies inside its utterance, else tile_hi
        while (tile < tile_hi) {
            const PTileRef r = tile_ref(tile);
            if (r.t0 < r.L) break;
            ++tile;
        }
        return tile;
    };
    int tile = next_valid(tile_lo);
    if (tile >= tile_hi) return;

    const unsigned short* const xbase = static_cast<const unsigned short*>(a.x);
    unsigned short* const ybase = static_cast<unsigned short*>(a.y);
    // the X rows of a tile start at time t0 - H2 - H1; an INTERIOR tile has all its buffer rows inside the utterance
    auto interior = [&](const PTileRef& r) { return r.t0 - H2 - H1 >= 0 && r.t0 - H2 - H1 + T::ROWS_BUF <= r.L; };

    // ---------------- per-lane constants (a handful: everything else is an immediate or an SGPR offset) -------------------
    const int rowbase0 = wn * (N1 / WN) + l31;                        // this lane's column of the wave tile's block 0 = tile row
    const unsigned b_row = (unsigned)rowbase0 * P;                      // its byte offset in a tile buffer
    const unsigned a_voff = (unsigned)((wm * MR) * 64 + lane) * 16;     // A fragment: lane's bytes inside a k-step's [MB][64][16 B]
    const unsigned e_voff = (unsigned)((wn * (N1 / WN) + l31) * P + (wm * (C / T::WM) + 8 * lh) * 2);  // epilogue-2 chunk of unit 0
    const unsigned l_voff = (unsigned)(wave * 1024 + lane * 16);        // DMA / LeakyReLU unit 0: this lane's 16 bytes in a buffer
    // DMA instruction j of this wave covers tile rows RPD*(wave + NWAVES*j) ..: lane l writes PHYSICAL slot l % SPR of row
    // .. + l / SPR (the LDS image of a DMA is lane-linear) and reads the LOGICAL slot phys ^ swz(row) of that row (swizzle on
    // the source side, guide §5.4 rule 21).  swz(row) = row & 15 and the rows of consecutive j differ by 16: one constant.
    const unsigned d_voff = (unsigned)(((RPD * wave + lane / SPR) * C + (((lane % SPR) ^ ((RPD * wave + lane / SPR) & 15)) * 8)) * 2);
    static_assert(RPD * NWAVES == 16, "source swizzle constant per lane");

    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, (int)(2 * T::CONV_BYTES), 0x00020000);

    // ---------------- LDS-DMA of a tile's X rows (raw bf16) into a buffer ---------------------------------------------------
    auto dma_interior = [&](const PTileRef& r, unsigned nbuf_off, int j) {  // j static or wave-uniform
        const unsigned short* sb = xbase + ((size_t)r.b * Lp + (r.t0 - H2 - H1)) * C + (size_t)j * (RPD * NWAVES * C);
        glds16_saddr(d_voff, sb, sgpr_i(lds_addr_of(lds) + nbuf_off + wave * 1024 + j * (NWAVES * 1024)));
    };
    auto dma_edge = [&](const PTileRef& r, unsigned nbuf_off, int j) {  // rows outside the utterance: a clamped row (zeroed later)
        const int rowi = RPD * (wave + NWAVES * j);
        const int row = rowi + lane / SPR;
        int t = r.t0 - H2 - H1 + row;
        t = t < 0 ? 0 : (t >= r.L ? r.L - 1 : t);
        const unsigned short* src = xbase + ((size_t)r.b * Lp + t) * C + (((lane % SPR) ^ (row & 15)) * 8);
        glds16_asm(src, sgpr_i(lds_addr_of(lds) + nbuf_off + rowi * P));
    };
    // in-place LeakyReLU of unit j: the 16 bytes this lane DMA'd (EDGE: zero padding outside the utterance first)
    auto act2 = [](unsigned u) { return lrelu01_pack(bf16_lo(u), bf16_hi(u)); };  // LRELU_SLOPE, model.py:5,46
    auto lrelu_at = [&](unsigned addr, bool ok) {
        uint4* p = reinterpret_cast<uint4*>(lds + addr);
        uint4 v = *p;
        v.x = ok ? v.x : 0u;
        v.y = ok ? v.y : 0u;
        v.z = ok ? v.z : 0u;
        v.w = ok ? v.w : 0u;
        v.x = act2(v.x);
        v.y = act2(v.y);
        v.z = act2(v.z);
        v.w = act2(v.w);
        *p = v;
    };

    // ---------------- accumulators, rings --------------------------------------------------------------------------------
    f32x16 accA[MR][NR], accB[MR][NR];  // c1 / c2 accumulators (c2's stay live through the next tile's c1 loop)
    f32x16 bblk[MR];
    bf16x8 af[RA][MR], bfr[2][NR];
    u32x4 rv[NE2];                       // residual rows of the tile whose epilogue 2 is pending

    auto load_bias = [&](const float* __restrict__ bias) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef float f32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const float* __restrict__ bp = bias + wm * (C / T::WM) + mr * 32 + 4 * lh;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(bp), q1 = *reinterpret_cast<const f32x4*>(bp + 8);
            const f32x4 q2 = *reinterpret_cast<const f32x4*>(bp + 16), q3 = *reinterpret_cast<const f32x4*>(bp + 24);
            const f32x8 lo = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7);
            bblk[mr] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        }
    };
    // A fragments of stream step g (0 .. 2*NQ - 1: c1's k-steps, then c2's; the stream wraps to the next tile's c1):
    // lane offset in a VGPR, the step's offset in an SGPR, the m-block as an immediate
    auto load_a = [&](int g, int slot) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_voff + mr * 1024, g * (MB * 1024), 0);
            af[slot][mr] = __builtin_bit_cast(bf16x8, v);
        }
    };
    // B fragments of one k-step: row (lane's row + tap*rate + 32*nr), slot (2*ks + lh) ^ (row & 15).  tapaddr / xs are this
    // tap's row address and swizzle term (computed once per block), the column block is an immediate offset.
    auto load_b = [&](unsigned tapaddr, unsigned xs, int ks, int par) {
        const unsigned addr = tapaddr + (xs ^ (unsigned)(ks << 5));
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) bfr[par][nr] = *reinterpret_cast<const bf16x8*>(lds + addr + nr * 32 * P);
    };
    auto tap_terms = [&](unsigned buf_off, int row_off, unsigned& tapaddr, unsigned& xs) {
        unsigned rb = b_row;
        opaque(rb);
        tapaddr = buf_off + rb + (unsigned)row_off * P;
        xs = ((((unsigned)(rowbase0 + row_off)) & 15u) ^ (unsigned)lh) << 4;
        opaque(xs);
    };
    auto swap_pair = [](unsigned& pd, unsigned& qd) {
        auto r = __builtin_amdgcn_permlane32_swap(pd, qd, false, false);
        pd = r[0];
        qd = r[1];
    };

    // ---------------- epilogue 2 of a finished tile: unit u = (mr, p, nr) -------------------------------------------------
    // unit u's chunk: row t0 + wn*(N1/WN) + nr*32 + l31, channels wm*(C/WM) + mr*32 + 16*p + 8*lh .. + 7 of its utterance:
    // lane offset (VGPR) + channel offset (immediate) + row offset of the tile and of nr (SGPR)
    auto resid_issue = [&](const PTileRef& r) {  // all of a tile's residual rows (x rows of its outputs) into rv[]
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xbase + (size_t)r.b * Lp * C), 0, r.L * P, 0x00020000);
#pragma unroll
        for (int u = 0; u < NE2; ++u) {
            const int nr = u % NR, p = (u / NR) % 2, mr = u / (2 * NR);
            rv[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, e_voff + (mr * 32 + p * 16) * 2, (r.t0 + nr * 32) * P, 0);  // rows >= L read as 0 (never stored)
        }
    };
    auto ep2_unit = [&](const PTileRef& r, auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
        constexpr int nr = u % NR, p = (u / NR) % 2, mr = u / (2 * NR);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(ybase + (size_t)r.b * Lp * C, 0, r.L * P, 0x00020000);
        unsigned qx = rv[u].x, qy = rv[u].y, qz = rv[u].z, qw = rv[u].w;
        swap_pair(qx, qz);  // 8 consecutive channels per lane -> the accumulator layout's 4 + 4
        swap_pair(qy, qw);
        // Both accumulator sets live in the accumulator half of the register file (2 x 128 = all 256 AGPRs): the VALU-visible
        // half is needed for the rings, the residual rows and the fillers' temporaries (with accB there the c1 blocks spilled,
        // and every reload drains the weight stream: scratch_load + s_waitcnt vmcnt(0)).  Read the 8 values explicitly.
        constexpr int r0 = 8 * p;
        float cv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) asm("v_accvgpr_read_b32 %0, %1" : "=v"(cv[e]) : "a"(accB[mr][nr][r0 + e]));
        const float v0 = bf16_lo(qx) + cv[0], v1 = bf16_hi(qx) + cv[1], v2 = bf16_lo(qy) + cv[2], v3 = bf16_hi(qy) + cv[3];
        const float v4 = bf16_lo(qz) + cv[4], v5 = bf16_hi(qz) + cv[5], v6 = bf16_lo(qw) + cv[6], v7 = bf16_hi(qw) + cv[7];
        unsigned p0 = pack_bf16x2(v0, v1), p1 = pack_bf16x2(v2, v3), q0 = pack_bf16x2(v4, v5), q1 = pack_bf16x2(v6, v7);
        swap_pair(p0, q0);
        swap_pair(p1, q1);
        unsigned off = e_voff + (mr * 32 + p * 16) * 2;
        // rows past the tile's NT2 outputs go nowhere (an offset outside the buffer: the store is dropped, as are rows >= L)
        if constexpr ((WN - 1) * (N1 / WN) + nr * 32 + 31 >= NT2) off = (wn * (N1 / WN) + nr * 32 + l31 < NT2) ? off : DROP;
        const u32x4 o = {p0, p1, q0, q1};
        if constexpr (PEXP & 8) asm volatile("" ::"v"(o));  // timing experiment: no store
        else __builtin_amdgcn_raw_buffer_store_b128(o, rs, off, (r.t0 + nr * 32) * P, 0);
    };

    // ---------------- round 6 (PEXP bit 5): the filler units cut into slices of <= 4 instructions, one per MFMA slot ------------------------------
    unsigned eq[4], epk[4];
    float ev[8];
    auto ep2_slice = [&](const PTileRef& r, auto u_tag, auto s_tag) {
        constexpr int u = decltype(u_tag)::value, S = decltype(s_tag)::value;
        constexpr int nr = u % NR, p = (u / NR) % 2, mr = u / (2 * NR), r0 = 8 * p;
        if constexpr (S == 0) {
            eq[0] = rv[u].x; eq[1] = rv[u].y; eq[2] = rv[u].z; eq[3] = rv[u].w;
            swap_pair(eq[0], eq[2]);
        } else if constexpr (S == 1) {
            swap_pair(eq[1], eq[3]);
        } else if constexpr (S >= 2 && S < 10) {
            asm("v_accvgpr_read_b32 %0, %1" : "=v"(ev[S - 2]) : "a"(accB[mr][nr][r0 + S - 2]));
        } else if constexpr (S >= 10 && S < 18) {
            constexpr int e = S - 10;
            ev[e] = ((e & 1) ? bf16_hi(eq[e / 2]) : bf16_lo(eq[e / 2])) + ev[e];
        } else if constexpr (S >= 18 && S < 22) {
            epk[S - 18] = pack_bf16x2(ev[2 * (S - 18)], ev[2 * (S - 18) + 1]);
        } else if constexpr (S == 22) {
            swap_pair(epk[0], epk[2]);
        } else if constexpr (S == 23) {
            swap_pair(epk[1], epk[3]);
        } else if constexpr (S == 24) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(ybase + (size_t)r.b * Lp * C, 0, r.L * P, 0x00020000);
            unsigned off = e_voff + (mr * 32 + p * 16) * 2;
            if constexpr ((WN - 1) * (N1 / WN) + nr * 32 + 31 >= NT2) off = (wn * (N1 / WN) + nr * 32 + l31 < NT2) ? off : DROP;
            const u32x4 o = {epk[0], epk[1], epk[2], epk[3]};
            if constexpr (PEXP & 8) asm volatile("" ::"v"(o));
            else __builtin_amdgcn_raw_buffer_store_b128(o, rs, off, (r.t0 + nr * 32) * P, 0);
        }
    };
    uint4 lv;
    float la[2];
    auto lrelu_slice = [&](unsigned addr, auto s_tag, auto edge_tag, const PTileRef& rn, int j) {
        constexpr int S = decltype(s_tag)::value;
        constexpr bool EDGE = decltype(edge_tag)::value;
        if constexpr (S == 0) {
            lv = *reinterpret_cast<const uint4*>(lds + addr);
        } else if constexpr (S == 7) {
            if constexpr (EDGE) {
                const int row = RPD * (wave + NWAVES * j) + lane / SPR;
                const bool ok = (unsigned)(rn.t0 - H2 - H1 + row) < (unsigned)rn.L;
                lv.x = ok ? lv.x : 0u; lv.y = ok ? lv.y : 0u; lv.z = ok ? lv.z : 0u; lv.w = ok ? lv.w : 0u;
            }
        } else if constexpr (S >= 8 && S < 24) {
            constexpr int c = (S - 8) / 4, st = (S - 8) % 4;
            unsigned& w = c == 0 ? lv.x : (c == 1 ? lv.y : (c == 2 ? lv.z : lv.w));
            if constexpr (st == 0) {
                la[0] = bf16_lo(w);
                la[1] = bf16_hi(w);
            } else if constexpr (st == 1) {
                la[0] = vmax_raw(la[0], vmul_raw(la[0], 0.1f));
            } else if constexpr (st == 2) {
                la[1] = vmax_raw(la[1], vmul_raw(la[1], 0.1f));
            } else {
                w = pack_bf16x2(la[0], la[1]);
            }
        } else if constexpr (S == 24) {
            *reinterpret_cast<uint4*>(lds + addr) = lv;
        }
    };

    // ---------------- one block of UB k-steps (= one tap) of the MFMA stream, with its fillers ------------------------------
    // PHASE 0: c1 over X (taps at rate DIL), accumulators accA;  PHASE 1: c2 over xt (rate 1), accumulators accB.
    // SIDE: 0 none | 1 QE epilogue-2 units starting at EU0 | 2 QL in-place LeakyReLU units from unit j0 (interior tile) |
    //       5 the same with the zero padding of an edge tile | 3 DMA burst of the next (interior) tile (last c1 block) |
    //       4 residual burst of this tile (last c2 block)
    auto block = [&](auto phase_tag, auto side_tag, auto eu0_tag, auto first_tag, int blk, unsigned buf_off, unsigned nbuf_off,
                     const PTileRef& rprev, const PTileRef& rcur, const PTileRef& rnext, int j0) {
        constexpr int PHASE = decltype(phase_tag)::value, SIDE = decltype(side_tag)::value, EU0 = decltype(eu0_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;  // the phase's first block: its first step starts from the bias block
        constexpr int RATE = PHASE == 0 ? DIL : 1;
        constexpr bool LAST = SIDE == 3 || SIDE == 4;
        __builtin_amdgcn_sched_barrier(0);
        const int g0 = PHASE * NQ + blk * UB;  // stream step of this block's first k-step
        unsigned tapaddr, xs, tapaddr_n = 0, xs_n = 0, laddr = 0;
        tap_terms(buf_off, blk * RATE, tapaddr, xs);
        if constexpr (!LAST) tap_terms(buf_off, (blk + 1) * RATE, tapaddr_n, xs_n);
        if constexpr (SIDE == 2 || SIDE == 5) {
            laddr = l_voff;
            opaque(laddr);
            laddr += nbuf_off + (unsigned)j0 * (NWAVES * 1024);
        }
        if constexpr (PEXP & 32) {
            // ---- round 6: the issue order written out.  Slot (i, k) = behind MFMA k of k-step i: the memory instruction due there (k < NR: the next step's B
            // fragment of column block k; k < NR + MR: the A fragment PA steps ahead) and one SLICE of a filler unit (<= 4 instructions whose operands were
            // produced at least 6 slots earlier), then a scheduling barrier: hipcc neither clumps the fillers in front of the step nor sinks a load to its use.
            static_for<UB>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
                static_for<MR * NR>([&](auto k_tag) {
                    constexpr int k = decltype(k_tag)::value, mr = k / NR, nr = k % NR, slot = i * (MR * NR) + k;
                    if constexpr (PHASE == 0)
                        accA[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i % RA][mr], bfr[i & 1][nr], (FIRST && i == 0) ? bblk[mr] : accA[mr][nr], 0, 0, 0);
                    else
                        accB[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i % RA][mr], bfr[i & 1][nr], (FIRST && i == 0) ? bblk[mr] : accB[mr][nr], 0, 0, 0);
                    if constexpr (k < NR) {
                        if constexpr (i + 1 < UB) bfr[(i + 1) & 1][k] = *reinterpret_cast<const bf16x8*>(lds + tapaddr + (xs ^ (unsigned)((i + 1) << 5)) + k * 32 * P);
                        else if constexpr (!LAST) bfr[(i + 1) & 1][k] = *reinterpret_cast<const bf16x8*>(lds + tapaddr_n + xs_n + k * 32 * P);
                    } else if constexpr (k < NR + MR) {
                        int ga = g0 + i + PA;
                        ga = ga >= 2 * NQ ? ga - 2 * NQ : ga;
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_voff + (k - NR) * 1024, ga * (MB * 1024), 0);
                        af[(i + PA) % RA][k - NR] = __builtin_bit_cast(bf16x8, v);
                    }
                    if constexpr (PEXP & 1) {
                    } else if constexpr (SIDE == 1) {
                        constexpr int U = EU0 + slot / 32, S = slot % 32;
                        if constexpr (U < NE2) ep2_slice(rprev, std::integral_constant<int, U>{}, std::integral_constant<int, S>{});
                    } else if constexpr (SIDE == 2 || SIDE == 5) {
                        constexpr int q = slot / 32, S = slot % 32;
                        if constexpr (q < QL) lrelu_slice(laddr + q * (NWAVES * 1024), std::integral_constant<int, S>{}, std::integral_constant<bool, SIDE == 5>{}, rnext, j0 + q);
                    } else if constexpr (SIDE == 3) {
                        constexpr int per = (NDW + UB - 1) / UB;
                        if constexpr (k >= MR * NR - per) {
                            constexpr int j = i * per + (k - (MR * NR - per));
                            if constexpr (j < NDW && !(PEXP & 2)) dma_interior(rnext, nbuf_off, j);
                        }
                    } else if constexpr (SIDE == 4) {
                        if constexpr (i == UB - 1 && k == MR * NR - 1 && !(PEXP & 2)) resid_issue(rcur);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        } else {
        static_for<UB>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            int ga = g0 + i + PA;
            ga = ga >= 2 * NQ ? ga - 2 * NQ : ga;
            load_a(ga, (i + PA) % RA);
            if constexpr (i + 1 < UB) load_b(tapaddr, xs, i + 1, (i + 1) & 1);
            else if constexpr (!LAST) load_b(tapaddr_n, xs_n, 0, (i + 1) & 1);  // no B fragment follows the phase's last step
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    if constexpr (PHASE == 0)
                        accA[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i % RA][mr], bfr[i & 1][nr], (FIRST && i == 0) ? bblk[mr] : accA[mr][nr], 0, 0, 0);
                    else
                        accB[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i % RA][mr], bfr[i & 1][nr], (FIRST && i == 0) ? bblk[mr] : accB[mr][nr], 0, 0, 0);
                }
            // ---- fillers of this k-step ----
            if constexpr (PEXP & 1) {
            } else if constexpr (SIDE == 1) {
                if constexpr (i % (UB / QE) == 0 && EU0 + i / (UB / QE) < NE2) ep2_unit(rprev, std::integral_constant<int, EU0 + i / (UB / QE)>{});
            } else if constexpr (SIDE == 2 || SIDE == 5) {
                if constexpr (i % (UB / QL) == 0) {
                    constexpr int q = i / (UB / QL);
                    bool ok = true;
                    if constexpr (SIDE == 5) {
                        const int row = RPD * (wave + NWAVES * (j0 + q)) + lane / SPR;
                        ok = (unsigned)(rnext.t0 - H2 - H1 + row) < (unsigned)rnext.L;
                    }
                    lrelu_at(laddr + q * (NWAVES * 1024), ok);
                }
            } else if constexpr (SIDE == 3) {
                constexpr int per = (NDW + UB - 1) / UB;
#pragma unroll
                for (int q = 0; q < per; ++q)
                    if (!(PEXP & 2) && i * per + q < NDW) dma_interior(rnext, nbuf_off, i * per + q);
            } else if constexpr (SIDE == 4) {
                if constexpr (i == UB - 1 && !(PEXP & 2)) resid_issue(rcur);
            }
            // ---- issue order of this k-step: every MFMA followed by its share of the step's memory instructions and fillers
            {
                // The next step's B fragments first (one after each of the step's first NR MFMAs: a full step = 256 cycles of
                // LDS latency cover; with ONE wave per SIMD nobody else fills a stall), then the A fragments PA steps ahead.
                constexpr int NM = MR * NR;
#pragma unroll
                for (int k = 0; k < NM; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    if constexpr (PEXP & 4) {
                        if (k >= 1 && k <= MR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        else if (k > MR && k <= MR + NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    } else {
                        if (k < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // DS read: a B fragment of the next step
                        else if (k < NR + MR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read: an A fragment PA steps ahead
                    }
                    if constexpr (SIDE == 1 || SIDE == 2 || SIDE == 5) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // filler VALU
                }
            }
        });
        }
        // keep the c2 accumulators in the accumulator half of the register file across blocks (hipcc moved them into arch
        // VGPRs for the c2 loop: 256 moves per tile, and MFMAs whose C/D share the A/B operands' register banks)
        if constexpr (PHASE == 1 && !(PEXP & 16)) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) asm volatile("" : "+a"(accB[mr][nr]));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;

    // epilogue 1: accA -> LeakyReLU(0.1) -> bf16, zero outside [0, L) -> xt tile (over the X tile)
    auto ep1 = [&](const PTileRef& r, unsigned buf_off) {
        unsigned char* const buf = lds + buf_off;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cb = wm * (C / T::WM) + mr * 32 + 16 * p;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const int row = wn * (N1 / WN) + nr * 32 + l31;
                    const int tt = r.t0 - H2 + row;
                    const bool ok = (unsigned)tt < (unsigned)r.L;
                    const int r0 = 8 * p;
                    unsigned p0 = lrelu01_pack(accA[mr][nr][r0 + 0], accA[mr][nr][r0 + 1]);
                    unsigned p1 = lrelu01_pack(accA[mr][nr][r0 + 2], accA[mr][nr][r0 + 3]);
                    unsigned q0 = lrelu01_pack(accA[mr][nr][r0 + 4], accA[mr][nr][r0 + 5]);
                    unsigned q1 = lrelu01_pack(accA[mr][nr][r0 + 6], accA[mr][nr][r0 + 7]);
                    if (!ok) p0 = p1 = q0 = q1 = 0u;  // c2's own zero padding applies to xt
                    swap_pair(p0, q0);
                    swap_pair(p1, q1);
                    const int slot = (cb >> 3) + lh;
                    *reinterpret_cast<uint4*>(buf + row * P + ((slot ^ swz_of<SPR>(row)) << 4)) = make_uint4(p0, p1, q0, q1);
                }
            }
        }
        // rows N1 .. N1 + 2*H2 - 1 are only read by the discarded output columns: keep them finite
        for (int u = tid; u < 2 * H2 * SPR; u += T::THREADS) {
            const int row = N1 + u / SPR, c = u % SPR;
            *reinterpret_cast<uint4*>(buf + row * P + (c << 4)) = make_uint4(0u, 0u, 0u, 0u);
        }
    };

    // =========================== prologue: the run's first tile, nothing to hide it behind ================================
    PTileRef rcur = tile_ref(tile), rprev = rcur, rnext = rcur;
    unsigned buf_off = 0, nbuf_off = T::BUF_BYTES;
    {
#pragma unroll 1
        for (int j = 0; j < NDW; ++j) dma_edge(rcur, buf_off, j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int j = 0; j < NDW; ++j) {
            const int row = RPD * (wave + NWAVES * j) + lane / SPR;
            lrelu_at(buf_off + l_voff + j * (NWAVES * 1024), (unsigned)(rcur.t0 - H2 - H1 + row) < (unsigned)rcur.L);
        }
        load_bias(a.bias);
#pragma unroll
        for (int s = 0; s < PA; ++s) load_a(s, s % RA);
    }
    __syncthreads();
    bool have_prev = false;
#if VTTS_TIMELINE
    unsigned long long tl_c1 = 0, tl_e1 = 0, tl_c2 = 0, tl_n = 0, tl_t = __builtin_amdgcn_s_memtime();
    const unsigned long long tl_start = tl_t;
#define PTL(acc)                                                   \
    do {                                                           \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
        acc += n_ - tl_t;                                          \
        tl_t = n_;                                                 \
    } while (0)
#else
#define PTL(acc) do { } while (0)
#endif

#pragma unroll 1
    for (;;) {
        const int ntile = next_valid(tile + 1);
        const bool have_next = ntile < tile_hi;
        rnext = have_next ? tile_ref(ntile) : rcur;  // no next tile: stage this one again (harmless, keeps the stream uniform)
        const bool nint = interior(rnext);

        // -------- c1 loop: X(i) in buf -> accA;  fillers: epilogue 2 of tile i-1; DMA of tile i+1 in the last block --------
        {
            unsigned ta, xs;
            tap_terms(buf_off, 0, ta, xs);
            load_b(ta, xs, 0, 0);
        }
        if (have_prev) {
            // the blocks that carry epilogue-2 units are unrolled (their units index registers statically)
            static_for<NBE>([&](auto b_tag) {
                constexpr int b = decltype(b_tag)::value;
                block(I0{}, I1{}, std::integral_constant<int, b * QE>{}, std::integral_constant<bool, b == 0>{}, b, buf_off, nbuf_off, rprev, rcur, rnext, 0);
            });
        } else {
            block(I0{}, I0{}, I0{}, std::true_type{}, 0, buf_off, nbuf_off, rprev, rcur, rnext, 0);
#pragma unroll 1
            for (int b = 1; b < NBE; ++b) block(I0{}, I0{}, I0{}, std::false_type{}, b, buf_off, nbuf_off, rprev, rcur, rnext, 0);
        }
#pragma unroll 1
        for (int b = NBE; b < NBLK - 1; ++b) block(I0{}, I0{}, I0{}, std::false_type{}, b, buf_off, nbuf_off, rprev, rcur, rnext, 0);
        load_bias(a.bias + C);  // c2's bias block, ahead of the DMA burst in the load queue
        if (nint) {
            block(I0{}, I3{}, I0{}, std::false_type{}, NBLK - 1, buf_off, nbuf_off, rprev, rcur, rnext, 0);
        } else {  // the next tile touches an utterance edge (2 tiles per utterance): clamped addresses, issued after the loop
            block(I0{}, I4{}, I0{}, std::false_type{}, NBLK - 1, buf_off, nbuf_off, rprev, rcur, rnext, -1);
#pragma unroll 1
            for (int j = 0; j < NDW; ++j) dma_edge(rnext, nbuf_off, j);
        }
        PTL(tl_c1);
        __syncthreads();  // every wave is done reading X(i)

        // -------- epilogue 1 (exposed) --------
        ep1(rcur, buf_off);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of X(i+1) have landed (and everything older)
        __syncthreads();  // xt(i) written
        PTL(tl_e1);

        // -------- c2 loop: xt(i) in buf -> accB;  fillers: LeakyReLU of X(i+1) in place; residual burst in the last block ----
        {
            unsigned ta, xs;
            tap_terms(buf_off, 0, ta, xs);
            load_b(ta, xs, 0, 0);
        }
        if (nint) {
            block(I1{}, I2{}, I0{}, std::true_type{}, 0, buf_off, nbuf_off, rprev, rcur, rnext, 0);
#pragma unroll 1
            for (int b = 1; b < NBL; ++b) block(I1{}, I2{}, I0{}, std::false_type{}, b, buf_off, nbuf_off, rprev, rcur, rnext, b * QL);
        } else {
            block(I1{}, I5{}, I0{}, std::true_type{}, 0, buf_off, nbuf_off, rprev, rcur, rnext, 0);
#pragma unroll 1
            for (int b = 1; b < NBL; ++b) block(I1{}, I5{}, I0{}, std::false_type{}, b, buf_off, nbuf_off, rprev, rcur, rnext, b * QL);
        }
#pragma unroll 1
        for (int b = NBL; b < NBLK - 1; ++b) block(I1{}, I0{}, I0{}, std::false_type{}, b, buf_off, nbuf_off, rprev, rcur, rnext, 0);
        load_bias(a.bias);  // the next tile's c1 bias block
        block(I1{}, I4{}, I0{}, std::false_type{}, NBLK - 1, buf_off, nbuf_off, rprev, rcur, rnext, 0);
        PTL(tl_c2);
        __syncthreads();  // every wave is done with xt(i); X(i+1) is activated
#if VTTS_TIMELINE
        ++tl_n;
#endif

        rprev = rcur;
        have_prev = true;
        if (!have_next) break;
        rcur = rnext;
        tile = ntile;
        const unsigned t_ = buf_off;
        buf_off = nbuf_off;
        nbuf_off = t_;
    }
    // =========================== drain: epilogue 2 of the run's last tile ====================================================
    static_for<NE2>([&](auto u_tag) { ep2_unit(rprev, u_tag); });
#if VTTS_TIMELINE
    if (a.dbg && tid == 0) {
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 16;
        d[0] = tl_c1;
        d[1] = tl_e1;
        d[2] = tl_c2;
        d[3] = tl_n;
        d[4] = __builtin_amdgcn_s_memtime() - tl_start;
        d[6] = 1;
    }
#endif
}

// ---- tile table -------------------------------------------------------------------------------------
//                                                     C   KS  DIL  N1  WM WN PA QE QL
template <int KS, int DIL> struct P128;
#ifndef PEXP_PA  // round 6: A-fragment look-ahead in k-steps (3 = a 4-slot ring; 7 = an 8-slot ring: stores and slow loads issued behind a fragment request do not delay it)
#define PEXP_PA 3
#endif
template <int DIL> struct P128<11, DIL> { using type = PTile<128, 11, DIL, 256, 2, 2, PEXP_PA, 2, 2>; };
template <int DIL> struct P128<7, DIL> { using type = PTile<128, 7, DIL, 256, 2, 2, 3, 4, 4>; };

static int g_num_cus = 0;
static int num_cus() {
    if (!g_num_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

template <class T>
static hipError_t launch_p(const BConvArgs& a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_p_bf16_k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tpu = (a.L + T::NT2 - 1) / T::NT2;
    const int ntiles = tpu * a.B;
    int grid = num_cus();
    if (grid > ntiles) grid = ntiles;
    const int run = (ntiles + grid - 1) / grid;
    grid = (ntiles + run - 1) / run;
    hipLaunchKernelGGL(resblock_pair_p_bf16_k<T>, dim3(grid), dim3(T::THREADS), T::LDS_BYTES, s, a, tpu, ntiles, run);
    return hipGetLastError();
}

bool pair_p_bf16_supported(int C, int K, int dil, const BConvArgs& a) {
    if (a.acc_add != 0 || a.div != 1.0f || a.slope_out != 1.0f) return false;  // the MRF read-modify-write pairs: second generation
    return C == 128 && (K == 7 || K == 11) && (dil == 1 || dil == 3 || dil == 5);
}

hipError_t launch_pair_p_bf16(int C, int K, const BConvArgs& a, hipStream_t s) {
#ifdef VTTS_P_ONLY_K11_D3  // kernel development: one instantiation (compile time)
    if (C == 128 && K == 11 && a.dil == 3) return launch_p<typename P128<11, 3>::type>(a, s);
    return hipErrorInvalidValue;
#else
    if (C == 128 && K == 11) switch (a.dil) {
            case 1: return launch_p<typename P128<11, 1>::type>(a, s);
            case 3: return launch_p<typename P128<11, 3>::type>(a, s);
            case 5: return launch_p<typename P128<11, 5>::type>(a, s);
        }
    if (C == 128 && K == 7) switch (a.dil) {
            case 1: return launch_p<typename P128<7, 1>::type>(a, s);
            case 3: return launch_p<typename P128<7, 3>::type>(a, s);
            case 5: return launch_p<typename P128<7, 5>::type>(a, s);
        }
    return hipErrorInvalidValue;
#endif
}

const char* pair_p_kernel_name(int C, int K) {
    static thread_local char buf[96];
    snprintf(buf, sizeof(buf), "resblock_pair_p_bf16_k<PTile<%d, %d,", C, K);
    return buf;
}

}  // namespace vtts
