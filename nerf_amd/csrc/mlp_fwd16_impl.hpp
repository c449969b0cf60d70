// K2, bf16 inference form for the 256-wide skip models (depth 4 or 8, encoders of at most 64
// slots): the same fused encode -> trunk -> density head -> colour head as mlp_fwd_impl.hpp, same
// transposed orientation (weights = MFMA A operand from the LDS ring, samples = B columns,
// activations never leave the register file), but built on v_mfma_f32_16x16x32_bf16.
//
// Why a second shape: the chip holds a higher clock under 16x16x32 than under 32x32x16 at equal
// cycles per FLOP, and the 16-row output tile wastes half as much on the 3-row colour layer.
//
// Lane geometry.  A 16x16 accumulator has its sample on lane % 16 and rows 4 (lane / 16) + r in
// its four registers.  A wavefront still owns 32 samples, as two blocks of 16 columns: lane l
// carries sample 16 s + l % 16 of block s = 0, 1, and lane group q = l / 16 carries, for BOTH of
// its samples, inputs 8 q .. 8 q + 7 of every k-unit (32 input features) and rows 4 q .. 4 q + 3
// of every 16-row output tile.  One weight unit (1 KiB) feeds two MFMAs, one per block.  Two
// adjacent output tiles, converted to bf16, are in place the next layer's B operand of one
// k-unit (k order: chain_unit_feature(NERFX_LAYOUT_M16, ...), baked into the stream by the packer).
//
// A layer computes four output tiles (64 features) at a time: 8 accumulators of 4 registers.
//
// Phases.  A workgroup is 8 wavefronts, two per SIMD (waves w and w + 4 share one).  Waves 0..3
// (group 0) and waves 4..7 (group 1) run the same program half a step apart: while one group is
// inside an MFMA run, the other converts its results, refills the weight ring and prepares its
// next run, so the matrix pipe of every SIMD always has one wave feeding it and the other wave's
// VALU / LDS-DMA work fills the issue slots between the MFMAs.  Every run is bracketed by two
// workgroup barriers (inside the hand-scheduled blocks); group 1 meets one extra barrier at the
// start and group 0 one at the end, which is all it takes to keep the groups in anti-phase.
//
// Weight ring.  The stream is consumed in SEGMENTS (one group of output tiles of one layer: its
// encoder units, then its chained units; 8, 32 or 40 units).  Segment s is read by group 0 in
// phase 2 s and by group 1 in phase 2 s + 1; after that group 1 refills the first half of its
// slot and, one phase later, group 0 the second half (each wave a quarter of a half), with the
// stream units that lie one ring length ahead.  A wave waits for its own loads before its next
// barrier, so what was requested in phase p is visible to everybody from phase p + 1.  A ring of
// two maximal segments (80 units) is what this needs: every unit of segment n has then been
// requested by phase 2 n - 2 (its first units, which are read ahead before the barrier) resp.
// 2 n - 1; the ring is 112 units for the static mode below, which requests everything 32 units earlier.
#pragma once
#include "mlp_fwd_impl.hpp"

namespace nerfx {

#include "mlp_runs_m16.inc"

constexpr int M16 = NERFX_LAYOUT_M16;
constexpr int M16_KU = 8;                       // k-units of a 256-wide activation
constexpr int M16_ENC_KU = 2;                   // k-units of an encoder (64 slots)
constexpr int M16_STASH_BYTES_PER_WAVE = M16_ENC_KU * 2 * UNIT_BYTES;
static_assert(M16_ENC_KU == enc_block(M16), "encoder segment size");

static_assert(M16_RING_UNITS >= 80, "at least two segments of 40 units");
static_assert(M16_RING_UNITS % PIECE_UNITS == 0, "a piece never straddles the end of the ring");

struct PhasedRing {
    static constexpr int CHUNK = 4;              // units per refill chunk: one LDS base (M0), four loads
    const char* src;             // stream base (wave-uniform; lanes add their own 32-bit offset)
    lds_bytes_t lds;             // ring base
    uint32_t n_units;            // stream length of this launch, in units
    uint32_t cons, cons_stream;  // ring / stream unit index of the current segment's first unit
    uint32_t prev_cons, prev_stream, prev_units;   // group 0: the slot whose second half is still to refill
    uint32_t fill_ring, fill_stream, fill_chunks;  // this group's refill duty of the current phase, in chunks
    int wave, lane, group, gw;   // group = wave / 4, gw = wave % 4

    __device__ __forceinline__ static uint32_t wrap_ring(uint32_t u) { return u >= M16_RING_UNITS ? u - M16_RING_UNITS : u; }
    __device__ __forceinline__ uint32_t wrap_stream(uint32_t u) const { return u >= n_units ? u - n_units : u; }
    // Four consecutive units (4-aligned in ring and stream, so they never straddle either end:
    // segments start at multiples of 8 in both) with ONE LDS base: the instruction offset moves
    // the global and the LDS address alike.  LDS-DMA instructions of a wave that each carry their
    // own base (M0) issue a full L2 round trip apart; with a shared base they go back to back.
    __device__ __forceinline__ void load_chunk(uint32_t ring_unit, uint32_t stream_unit) {
        const char* g = src + (size_t)wrap_stream(stream_unit) * UNIT_BYTES + (uint32_t)(lane * 16);
        lds_bytes_t l = lds + wrap_ring(ring_unit) * UNIT_BYTES;
        const auto gp = (const void __attribute__((address_space(1)))*)g;
        const auto lp = (void __attribute__((address_space(3)))*)l;
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 0 * UNIT_BYTES, 0);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 1 * UNIT_BYTES, 0);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 2 * UNIT_BYTES, 0);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 3 * UNIT_BYTES, 0);
    }
    __device__ __forceinline__ void init(const char* s, lds_bytes_t l, uint32_t nu, int w, int ln) {
        src = s; lds = l; n_units = nu; wave = w; lane = ln;
        group = w >> 2; gw = w & 3;
        cons = 0; cons_stream = 0; prev_cons = 0; prev_stream = 0; prev_units = 0;
        fill_ring = 0; fill_stream = 0; fill_chunks = 0;
        constexpr int CHUNKS = M16_RING_UNITS / CHUNK;
        for (int k = w; k < CHUNKS; k += RingBf16::WAVES) load_chunk(k * CHUNK, k * CHUNK);
    }
    // Static mode (stream length = whole ring lengths): at the top of every sample block the ring is
    // in the same state, which is written down here as constants -- consumption back at unit 0 of
    // ring and stream, and group 0 still owing the second half of the slot of the block's last
    // segment (LAST_UNITS long, at the end of the stream).  Everything the block's phases derive
    // from it (piece addresses, refill sources and targets, wrap-arounds) then folds at compile
    // time.  In a workgroup's first block nothing is owed yet; refilling that half slot anyway
    // rewrites ring units with the bytes the initial fill has put there.
    template <uint32_t STREAM_UNITS, uint32_t LAST_UNITS>
    __device__ __forceinline__ void restart() {
        static_assert(STREAM_UNITS % M16_RING_UNITS == 0 && STREAM_UNITS >= 2 * M16_RING_UNITS, "whole ring lengths");
        n_units = STREAM_UNITS;
        cons = 0; cons_stream = 0;
        prev_units = LAST_UNITS; prev_stream = STREAM_UNITS - LAST_UNITS; prev_cons = (STREAM_UNITS - LAST_UNITS) % M16_RING_UNITS;
    }
    // This wave's loads have landed (required before any barrier that publishes them).
    static __device__ __forceinline__ void landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    // LDS byte address, for this lane, of unit `rel` (a multiple of PIECE_UNITS) of the current segment.
    __device__ __forceinline__ uint32_t piece_ptr(int rel) const {
        return (uint32_t)(uintptr_t)(lds + wrap_ring(cons + rel) * UNIT_BYTES + lane * 16);
    }
    // Behind the closing barrier of this wave's run over a segment of U units: work out the refill
    // duty of this wave's group, then move on.  Group 1: first half of the slot just left (group 0
    // left it a phase ago).  Group 0: second half of the slot BEFORE it (group 1 is still reading
    // the one just left).  Halves are cut at chunk boundaries; wave gw issues chunks gw, gw + 4, ...
    // of its group's half through refill(round).
    template <int U>
    __device__ __forceinline__ void finish() {
        static_assert(U % 8 == 0, "segments are whole pieces");
        if (group) {
            fill_chunks = (U / CHUNK + 1) / 2;
            fill_ring = cons;
            fill_stream = cons_stream + M16_RING_UNITS;
        } else {
            const uint32_t first = (prev_units / CHUNK + 1) / 2;       // chunks of the other group's half
            fill_chunks = prev_units / CHUNK - first;
            fill_ring = prev_cons + first * CHUNK;
            fill_stream = prev_stream + M16_RING_UNITS + first * CHUNK;
            prev_cons = cons; prev_stream = cons_stream; prev_units = U;
        }
        cons = wrap_ring(cons + U);
        cons_stream = wrap_stream(cons_stream + U);
    }
    __device__ __forceinline__ void refill(uint32_t round) {
#ifdef NERFX_DIAG_NO_REFILL                      // timing diagnostic (wrong results): what do the refill loads cost?
        return;
#endif
        const uint32_t k = gw + 4 * round;
        if (k < fill_chunks) load_chunk(fill_ring + k * CHUNK, fill_stream + k * CHUNK);
    }
};

using Frag16 = bf16x8;
struct Act16 { Frag16 f[M16_KU][2]; };          // one layer's activations: [k-unit][sample block]

// B operand of encoder k-unit u for lane group q: slots 32 u + 8 q .. + 7 = pair-slots
// 16 u + 4 q .. + 3 of the encoder's coefficient table (see encode_unit in mlp_fwd_impl.hpp).
__device__ __forceinline__ Frag16 encode_unit16(lds_floats_t table, bool positional, int u, int q, float sx, float sy, float sz,
                                                float x, float y, float z) {
    using P = Prec<NERFX_PREC_BF16>;
    const lds_floats_t c0 = table + 4 * (16 * u + 4 * q);
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 c = NERFX_LDS_VEC(f32x4, c0 + 4 * i);
        const float angle = fmaf(sz, c[2], fmaf(sy, c[1], __fmul_rn(sx, c[0])));
        P::sincos_pair(angle, v[2 * i], v[2 * i + 1]);
    }
    if (positional && u == 0 && q == 0) { v[0] = x; v[1] = y; v[2] = z; v[3] = 0.0f; }
    return P::from_floats(v);
}

// Two adjacent 16-row tiles of one sample block -> the B operand of the k-unit they form.
template <bool RELU>
__device__ __forceinline__ Frag16 m16_from_acc(const f32x4& lo, const f32x4& hi) {
    using P = Prec<NERFX_PREC_BF16>;
    using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
    const u32x4 p = {P::pack2(lo[0], lo[1]), P::pack2(lo[2], lo[3]), P::pack2(hi[0], hi[1]), P::pack2(hi[2], hi[3])};
    const Frag16 f = __builtin_bit_cast(Frag16, p);
    return RELU ? P::relu(f) : f;
}

// Activation of an fp32 accumulator that comes out of a run's asm block.  fmaxf on such a value is
// compiled as two v_max_f32, the first only to quiet a signalling NaN, which an accumulator never is:
// LeakyReLU is written out (same values, one instruction fewer); ReLU is an integer max (activate).
template <int ACT>
__device__ __forceinline__ float m16_activate(float x) {
    if constexpr (ACT == NERFX_ACT_LEAKY_RELU) {
        float y;                                                   // max(x, 0.01f * x)
        asm("v_mul_f32 %0, 0x3c23d70a, %1\n\tv_max_f32 %0, %1, %0" : "=&v"(y) : "v"(x));
        return y;
    } else {
        return activate<NERFX_PREC_BF16, ACT>(x);
    }
}

enum { M16_ENC_NONE = 0, M16_ENC_STASH = 1, M16_ENC_REGS = 2 };

// One Linear layer of NT 16-row output tiles on [encoder k-units | CK chained k-units].
//   ENC: where the encoder segment's B operands come from (the wave's LDS stash, or `encr`)
//   SIGMA: also accumulate the density head's dot product on the fp32 activations (per block)
// Each group of four tiles is one ring segment and one phase: prepare, run (between its two
// barriers), refill, convert.
template <int ACT, int NT, int CK, bool SIGMA, int ENC>
__device__ __forceinline__ void linear16(PhasedRing& ring, lds_floats_t bias, int q, const Act16& hin, Act16& hout,
                                         lds_floats_t sigw, float (&dot)[2], Stamps& st, uint32_t stash,
                                         const Frag16 (&encr)[M16_ENC_KU][2]) {
    static_assert(NT % 4 == 0 && (CK == 0 || CK == M16_KU), "full groups of four tiles on 0 or 256 chained inputs");
    static_assert(CK > 0 || ENC == M16_ENC_STASH, "an encoder-only layer reads the stash");
    constexpr int ENC_UNITS = ENC != M16_ENC_NONE ? 8 : 0, UNITS = ENC_UNITS + 4 * CK;
#pragma unroll
    for (int grp = 0; grp < NT / 4; ++grp) {
        // the biases are requested first and used last: their LDS round trip runs under the address
        // arithmetic (left alone, hipcc sinks the reads to their use and waits for them there)
        f32x4 bv[4], c[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t] = NERFX_LDS_VEC(f32x4, bias + 16 * (grp * 4 + t) + 4 * q);
        __builtin_amdgcn_sched_barrier(0);
        uint32_t p[UNITS / PIECE_UNITS];
#pragma unroll
        for (int k = 0; k < UNITS / PIECE_UNITS; ++k) p[k] = ring.piece_ptr(k * PIECE_UNITS);
        const uint32_t lane16 = (uint32_t)ring.lane * 16u;
        Frag16 a[M16_LOOKAHEAD];
        st.mark(ST_EPILOGUE);
        PhasedRing::landed();
        st.mark(ST_ACQUIRE);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) c[t][0] = c[t][1] = bv[t];
        // (the wave inside a run goes first on its SIMD: its partner's VALU work takes the slots
        // between the MFMAs)
        __builtin_amdgcn_s_setprio(3);
#define NERFX_C c[0][0], c[0][1], c[1][0], c[1][1], c[2][0], c[2][1], c[3][0], c[3][1], M16_A_ARGS(a)
#define NERFX_B(k0) hin.f[k0][0], hin.f[k0][1], hin.f[k0 + 1][0], hin.f[k0 + 1][1], hin.f[k0 + 2][0], hin.f[k0 + 2][1], hin.f[k0 + 3][0], hin.f[k0 + 3][1]
        Frag16 sb[2][2];                                            // stash operands as loaded (unused here)
        if constexpr (CK == 0) {
            m16_run8_stash_first_last(NERFX_C, p[0], sb[0][0], sb[0][1], sb[1][0], sb[1][1], stash, lane16, st);
        } else {
            constexpr int E = ENC_UNITS / PIECE_UNITS;                   // pieces in front of the chained ones
            if constexpr (ENC == M16_ENC_STASH) m16_run8_stash_first_out(NERFX_C, p[0], p[1], sb[0][0], sb[0][1], sb[1][0], sb[1][1], stash, lane16, st);
            if constexpr (ENC == M16_ENC_REGS) m16_run8_first_out(NERFX_C, p[0], p[1], encr[0][0], encr[0][1], encr[1][0], encr[1][1], st);
            if constexpr (ENC == M16_ENC_NONE) m16_run16_first_out(NERFX_C, p[0], p[1], p[2], NERFX_B(0), st);
            else m16_run16_out(NERFX_C, p[E], p[E + 1], p[E + 2], NERFX_B(0), st);
            m16_run16_last(NERFX_C, p[E + 2], p[E + 3], NERFX_B(4), st);
        }
#undef NERFX_C
#undef NERFX_B
        __builtin_amdgcn_s_setprio(0);
        st.mark(ST_RUN);
        ring.template finish<UNITS>();
        // epilogue: activation, density dot, conversion to the next layer's operands, with this
        // wave's refill loads spread between its four quarters
        constexpr bool packed_relu = ACT == NERFX_ACT_RELU && !SIGMA;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t == 0) ring.refill(0);
            if (t == 2) ring.refill(1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!packed_relu) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) c[t][s][r] = m16_activate<ACT>(c[t][s][r]);
            }
            if constexpr (SIGMA) {
                const f32x4 w = NERFX_LDS_VEC(f32x4, sigw + 16 * (grp * 4 + t) + 4 * q);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dot[s] = fmaf(c[t][s][r], w[r], dot[s]);
                // pin the sums here: hipcc otherwise sinks the whole chain (and with it the live fp32
                // activations) below the refill's branches, towards the density head
                asm volatile("" : "+v"(dot[0]), "+v"(dot[1]));
            }
#ifdef NERFX_DIAG_NO_EPILOGUE                    // timing diagnostic (wrong results): what does the conversion cost?
            if (t & 1) {
#pragma unroll
                for (int s = 0; s < 2; ++s) asm volatile("" : "=v"(hout.f[grp * 2 + t / 2][s]) : "v"(c[t - 1][s]), "v"(c[t][s]));
            }
#else
            if (t & 1) {
#pragma unroll
                for (int s = 0; s < 2; ++s) hout.f[grp * 2 + t / 2][s] = m16_from_acc<packed_relu>(c[t - 1][s], c[t][s]);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef NERFX_DIAG_SLEEP                         // timing diagnostic: is the phase between two runs on the critical path?
        __builtin_amdgcn_s_sleep(NERFX_DIAG_SLEEP);
#endif
        st.mark(ST_HEADS);
    }
}

// Layer 0: sixteen output tiles on the encoder's two k-units alone, as ONE segment of 32 units and
// one phase.  Nothing is chained in, so the registers that hold a chained layer's input carry the
// accumulators of all sixteen tiles; taken four tiles at a time like the other layers, its four
// runs of 16 MFMAs would each last as long as the partner's (much longer) phase between two runs.
template <int ACT>
__device__ __forceinline__ void linear16_first(PhasedRing& ring, lds_floats_t bias, int q, Act16& hout, Stamps& st, uint32_t stash) {
    constexpr int NT = 16, UNITS = NT * M16_ENC_KU;
    f32x4 c[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) c[t][0] = c[t][1] = NERFX_LDS_VEC(f32x4, bias + 16 * t + 4 * q);
    uint32_t p[UNITS / PIECE_UNITS];
#pragma unroll
    for (int k = 0; k < UNITS / PIECE_UNITS; ++k) p[k] = ring.piece_ptr(k * PIECE_UNITS);
    const uint32_t lane16 = (uint32_t)ring.lane * 16u;
    Frag16 a[M16_LOOKAHEAD], b[2][2];
    st.mark(ST_EPILOGUE);
    PhasedRing::landed();
    st.mark(ST_ACQUIRE);
    st.coarse(1);                                                   // (coarse marks: 1 = from a block's last run to the next block's first)
    __builtin_amdgcn_s_setprio(3);
#define NERFX_C4(t0) c[t0][0], c[t0][1], c[t0 + 1][0], c[t0 + 1][1], c[t0 + 2][0], c[t0 + 2][1], c[t0 + 3][0], c[t0 + 3][1], M16_A_ARGS(a)
    m16_run8_stash_first_out(NERFX_C4(0), p[0], p[1], b[0][0], b[0][1], b[1][0], b[1][1], stash, lane16, st);
    m16_run8_out(NERFX_C4(4), p[1], p[2], b[0][0], b[0][1], b[1][0], b[1][1], st);
    m16_run8_out(NERFX_C4(8), p[2], p[3], b[0][0], b[0][1], b[1][0], b[1][1], st);
    m16_run8_last(NERFX_C4(12), p[3], b[0][0], b[0][1], b[1][0], b[1][1], st);
#undef NERFX_C4
    __builtin_amdgcn_s_setprio(0);
    st.mark(ST_RUN);
    ring.template finish<UNITS>();
    constexpr bool packed_relu = ACT == NERFX_ACT_RELU;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t == 0) ring.refill(0);
        if (t == 8) ring.refill(1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!packed_relu) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) c[t][s][r] = m16_activate<ACT>(c[t][s][r]);
        }
        if (t & 1) {
#pragma unroll
            for (int s = 0; s < 2; ++s) hout.f[t / 2][s] = m16_from_acc<packed_relu>(c[t - 1][s], c[t][s]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    st.mark(ST_HEADS);
}

// MODE: 0 = ring positions tracked at run time (any stream length); static mode (see
// PhasedRing::restart): 1 = up to the density head, 2 = with the colour layers.
enum { M16_DYNAMIC = 0, M16_STATIC_DENSITY = 1, M16_STATIC_COLOUR = 2 };
template <int ACT, int HD, int MODE>
__global__ __launch_bounds__(64 * RingBf16::WAVES, RingBf16::WAVES / 4) void mlp_fwd16_kernel(const MlpArgs a) {
    using RG = RingBf16;
    using P = Prec<NERFX_PREC_BF16>;
    constexpr int WAVES = RG::WAVES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) float* cst = (__attribute__((address_space(3))) float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, col = lane & 15;

    // LDS: [const block | per-wave encoding stash | weight ring]
    const lds_bytes_t stash_base = (lds_bytes_t)smem + a.const_floats * (int)sizeof(float);
    const lds_bytes_t my_stash = stash_base + wave * M16_STASH_BYTES_PER_WAVE + lane * 16;
    PhasedRing ring;
    ring.init(a.stream, stash_base + WAVES * M16_STASH_BYTES_PER_WAVE, a.n_pieces * PIECE_UNITS, wave, lane);
    load_const_block<64 * WAVES>(a, cst, tid);
    PhasedRing::landed();                                           // this wave's share of the initial fill
    __syncthreads();
#ifdef NERFX_DIAG_SOLO                           // timing diagnostic (wrong results): group 0 alone, how long is a run by itself?
    if (ring.group == 1) return;
#endif
    if (ring.group == 1) __builtin_amdgcn_s_barrier();              // half a step behind group 0, from here on

    const lds_floats_t sigw = cst + a.sigw_off;
    const bool colour = MODE == M16_DYNAMIC ? a.rgb != nullptr : MODE == M16_STATIC_COLOUR;
    constexpr int RGB2_UNITS = 8 + (MODE == M16_STATIC_COLOUR ? M16_STATIC_PAD : 0);   // the padding rides with the last segment
    constexpr int LAST = 2 * HD - 1;                                // last trunk layer feeds the density head
    const bool x_positional = a.ex_kind == NERFX_ENC_POSITIONAL, d_positional = a.ed_kind == NERFX_ENC_POSITIONAL;
    // coordinates as the pair tables want them, in revolutions: v / 2 pi (positional), 2 pi v / 2 pi (Fourier)
    const float x_scale = x_positional ? P::ANGLE_UNIT : 1.0f, d_scale = d_positional ? P::ANGLE_UNIT : 1.0f;
    // samples of a ray are consecutive points (ray.py:132): with a whole number of 32-sample blocks per
    // ray, the block of a wave never spans two rays
    const bool one_ray = a.t != nullptr && a.n_samples % 32 == 0;
    // An x encoder with 17 .. 20 pair-slots (PE(3, 6): 2 raw + 18) uses, of its second k-unit, only
    // the first four pair-slots, which are lane group 0's: the groups then evaluate one of them each.
    const lds_bytes_t wave_stash = stash_base + wave * M16_STASH_BYTES_PER_WAVE;
    const int x_pairs = (enc_raw_slots(a.ex_kind) + (x_positional ? 6 : 2) * a.ex_n) / 2;
    const bool x_tail_spread = x_pairs > 16 && x_pairs <= 20;
    if (x_tail_spread) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            *reinterpret_cast<__attribute__((address_space(3))) Frag16*>(my_stash + (2 + s) * UNIT_BYTES) = __builtin_bit_cast(Frag16, s16x8(0));
    }

    // A block's inputs (sample position along the ray, origin, direction) are requested one block
    // ahead, right after the previous block's have been unpacked: their HBM round trip then runs
    // under a whole block of work instead of in front of the first layer.
    struct Inputs { float t[2], o[2][3], d[2][3]; };
    auto request = [&](uint32_t blk, Inputs& in) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t m = blk * (32 * WAVES) + wave * 32 + 16 * s + col;
            const uint32_t mc = m < a.n_points ? m : a.n_points - 1;
            const size_t r = a.t ? mc / a.n_samples : mc;           // ray.py:132 / :226: one origin / direction per ray
            in.t[s] = a.t ? a.t[mc] : 0.0f;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { in.o[s][ax] = a.ro[3 * r + ax]; in.d[s][ax] = a.rd[3 * r + ax]; }
        }
    };
    Inputs next;
    request(blockIdx.x < a.n_blocks ? blockIdx.x : 0, next);

    Stamps st;
    st.start();
    for (uint32_t blk = blockIdx.x; blk < a.n_blocks; blk += gridDim.x) {
        st.phase = 0;
        if constexpr (MODE == M16_STATIC_DENSITY) ring.template restart<m16_trunk_units(HD), 32>();
        if constexpr (MODE == M16_STATIC_COLOUR) ring.template restart<m16_total_units(HD) + M16_STATIC_PAD, RGB2_UNITS>();
        // ---- this lane's two samples (all four lane groups hold the same two) ----------
        uint32_t m[2];
        bool valid[2], poisoned[2];
        float px[2], py[2], pz[2], dx[2], dy[2], dz[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            m[s] = blk * (32 * WAVES) + wave * 32 + 16 * s + col;
            valid[s] = m[s] < a.n_points;
            dx[s] = next.d[s][0]; dy[s] = next.d[s][1]; dz[s] = next.d[s][2];
            if (a.t) {
                px[s] = __fadd_rn(next.o[s][0], __fmul_rn(dx[s], next.t[s]));
                py[s] = __fadd_rn(next.o[s][1], __fmul_rn(dy[s], next.t[s]));
                pz[s] = __fadd_rn(next.o[s][2], __fmul_rn(dz[s], next.t[s]));
            } else {
                px[s] = next.o[s][0]; py[s] = next.o[s][1]; pz[s] = next.o[s][2];
            }
            // A non-finite input poisons every feature in the reference (model.py:68-78).
            const float probe = (px[s] + py[s] + pz[s]) + (colour ? (dx[s] + dy[s] + dz[s]) : 0.0f);
            poisoned[s] = !(fabsf(probe) <= 3.0e38f);
        }
        // (the values above are in registers of their own before the next request overwrites `next`)
        asm volatile("" : "+v"(px[0]), "+v"(py[0]), "+v"(pz[0]), "+v"(dx[0]), "+v"(dy[0]), "+v"(dz[0]));
        asm volatile("" : "+v"(px[1]), "+v"(py[1]), "+v"(pz[1]), "+v"(dx[1]), "+v"(dy[1]), "+v"(dz[1]));
        {
            const uint32_t ahead = blk + gridDim.x;
            request(ahead < a.n_blocks ? ahead : blk, next);
        }
        st.mark(ST_PROLOGUE);
        // phi_x is consumed twice (layer 0 and the skip layer): its B operands are parked in the
        // wave's LDS stash, from where the encoder blocks read them.
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float sx = __fmul_rn(x_scale, px[s]), sy = __fmul_rn(x_scale, py[s]), sz = __fmul_rn(x_scale, pz[s]);
            *reinterpret_cast<__attribute__((address_space(3))) Frag16*>(my_stash + s * UNIT_BYTES) =
                encode_unit16(cst + a.tabx_off, x_positional, 0, q, sx, sy, sz, px[s], py[s], pz[s]);
            if (x_tail_spread) {
                // the second unit's used pair-slots (16 .. 19, all of lane group 0) one per lane group: the
                // same operations as encode_unit16, a quarter of them; the other groups' slots stay zero
                using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
                const f32x4 c = NERFX_LDS_VEC(f32x4, cst + a.tabx_off + 4 * (16 + q));
                float v0, v1;
                P::sincos_pair(fmaf(sz, c[2], fmaf(sy, c[1], __fmul_rn(sx, c[0]))), v0, v1);
                const bf16x2 packed = {(__bf16)v0, (__bf16)v1};
                *reinterpret_cast<__attribute__((address_space(3))) bf16x2*>(wave_stash + (2 + s) * UNIT_BYTES + col * 16 + 4 * q) = packed;
            } else {
                *reinterpret_cast<__attribute__((address_space(3))) Frag16*>(my_stash + (2 + s) * UNIT_BYTES) =
                    encode_unit16(cst + a.tabx_off, x_positional, 1, q, sx, sy, sz, px[s], py[s], pz[s]);
            }
        }
        // same wave reads it back: LDS is in order per wave
        const uint32_t stash = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(stash_base + wave * M16_STASH_BYTES_PER_WAVE));
        st.mark(ST_ENCODE);

        Act16 A, B;                                                 // ping-pong activations
        Frag16 encr[M16_ENC_KU][2] = {};
        float dot[2] = {0.0f, 0.0f};
        // ---- fc_1 then fc_2; fc_2's first layer sees [phi_x | h] (model.py:71-73).  Layer 0
        // writes A; the remaining 2 HD - 1 trunk layers alternate A -> B, B -> A and end in B.
        linear16_first<ACT>(ring, cst, q, A, st, stash);
#pragma unroll
        for (int li = 1; li + 1 <= LAST - 1; li += 2) {
            if (li == HD) linear16<ACT, 16, M16_KU, false, M16_ENC_STASH>(ring, cst + li * BIAS_STRIDE, q, A, B, sigw, dot, st, stash, encr);
            else linear16<ACT, 16, M16_KU, false, M16_ENC_NONE>(ring, cst + li * BIAS_STRIDE, q, A, B, sigw, dot, st, stash, encr);
            if (li + 1 == HD) linear16<ACT, 16, M16_KU, false, M16_ENC_STASH>(ring, cst + (li + 1) * BIAS_STRIDE, q, B, A, sigw, dot, st, stash, encr);
            else linear16<ACT, 16, M16_KU, false, M16_ENC_NONE>(ring, cst + (li + 1) * BIAS_STRIDE, q, B, A, sigw, dot, st, stash, encr);
        }
        static_assert(LAST != HD, "the skip layer is never the last trunk layer (HD >= 2)");
        linear16<ACT, 16, M16_KU, true, M16_ENC_NONE>(ring, cst + LAST * BIAS_STRIDE, q, A, B, sigw, dot, st, stash, encr);
        if (!colour) st.coarse(0);                                  // (coarse marks: 0 = from a block's first run to its last)
        // ---- density head (model.py:75, activations.py:33-35): lane groups 0 / 1 store blocks 0 / 1
        st.mark(ST_EPILOGUE);
        {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dot[s] += __shfl_xor(dot[s], 16, 64);
                dot[s] += __shfl_xor(dot[s], 32, 64);
            }
            const int s = q & 1;
            float sg = P::softplus(((s ? dot[1] : dot[0]) + cst[a.sigb_off]) - 1.0f);
            if (s ? poisoned[1] : poisoned[0]) sg = __builtin_nanf("");
            if (q < 2 && (s ? valid[1] : valid[0])) a.sigma[s ? m[1] : m[0]] = sg;
        }
        st.mark(ST_HEADS);
        if (colour) {
            // ---- colour head.  `features` is a Linear without activation (model.py:51, :76), so the
            // packer has folded it into rgb.0 (LayerPlan::folded): rgb.0 reads [phi_d | h] with h the
            // last trunk layer's output, B -> first half of A (model.py:53, :76).
            st.mark(ST_EPILOGUE);
            if (one_ray) {
                // The 32 samples of this wave's block lie on ONE ray, so all of them have the same phi_d:
                // its 32 pair-slots are evaluated once, one per lane (lanes 32..63 repeat 0..31), with
                // the operations of encode_unit16, and every lane picks the 16 bytes of its lane
                // group's slots up from the wave's stash (phi_x in it is dead after the skip layer;
                // LDS is in order per wave).
                const int pair = lane & 31;
                const float sx = __fmul_rn(d_scale, dx[0]), sy = __fmul_rn(d_scale, dy[0]), sz = __fmul_rn(d_scale, dz[0]);
                const f32x4 c = NERFX_LDS_VEC(f32x4, cst + a.tabd_off + 4 * pair);
                float v0, v1;
                P::sincos_pair(fmaf(sz, c[2], fmaf(sy, c[1], __fmul_rn(sx, c[0]))), v0, v1);
                if (d_positional && pair == 0) { v0 = dx[0]; v1 = dy[0]; }
                if (d_positional && pair == 1) { v0 = dz[0]; v1 = 0.0f; }
                using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
                const bf16x2 packed = {(__bf16)v0, (__bf16)v1};
                *reinterpret_cast<__attribute__((address_space(3))) bf16x2*>(wave_stash + 4 * pair) = packed;
#pragma unroll
                for (int u = 0; u < M16_ENC_KU; ++u)
                    encr[u][0] = encr[u][1] = NERFX_LDS_VEC(Frag16, wave_stash + 64 * u + 16 * q);
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float sx = __fmul_rn(d_scale, dx[s]), sy = __fmul_rn(d_scale, dy[s]), sz = __fmul_rn(d_scale, dz[s]);
#pragma unroll
                    for (int u = 0; u < M16_ENC_KU; ++u) encr[u][s] = encode_unit16(cst + a.tabd_off, d_positional, u, q, sx, sy, sz, dx[s], dy[s], dz[s]);
                }
            }
            st.mark(ST_ENCODE);
            linear16<ACT, 8, M16_KU, false, M16_ENC_REGS>(ring, cst + a.foldb_off, q, B, A, sigw, dot, st, stash, encr);
            constexpr int li = LAST + 3;                            // rgb.2's place among the layers (bias row)
            // ---- rgb.2 (one 16-row tile on 128 inputs: 4 units in one piece) + widened sigmoid
            // (model.py:54, activations.py:18-20)
            st.mark(ST_EPILOGUE);
            f32x4 out[2];
            out[0] = out[1] = NERFX_LDS_VEC(f32x4, cst + li * BIAS_STRIDE + 4 * q);
            const lds_bytes_t w0 = (lds_bytes_t)(uintptr_t)ring.piece_ptr(0);
            PhasedRing::landed();
            __builtin_amdgcn_s_barrier();                           // this segment's phase: compiler-scheduled, 8 MFMAs
            st.mark_wait(ST_ACQUIRE, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const Frag16 w = NERFX_LDS_VEC(bf16x8, w0 + k * UNIT_BYTES);
                out[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, A.f[k][0], out[0], 0, 0, 0);
                out[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, A.f[k][1], out[1], 0, 0, 0);
            }
            st.mark(ST_RUN);
            __builtin_amdgcn_s_barrier();
            st.mark_wait(ST_BARRIER, 1);
            st.coarse(0);
            ++st.phase;
            ring.template finish<RGB2_UNITS>();
            ring.refill(0);
            ring.refill(1);                                         // (group 0's duty here is half of rgb.0's last slot: 5 chunks)
            if (q == 0) {                                           // rows 0..2 of the tile sit in lane group 0
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (valid[s]) {
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            float v = P::widened_sigmoid(out[s][ch]);
                            if (poisoned[s]) v = __builtin_nanf("");
                            a.rgb[3 * (size_t)m[s] + ch] = v;
                        }
                    }
            }
        }
        st.mark(ST_HEADS);
    }
    if (ring.group == 0) __builtin_amdgcn_s_barrier();              // pairs with group 1's last closing barrier
    st.flush(a.stamps, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // no LDS-DMA may outlive the workgroup
}

template <int ACT, int HD, int MODE>
static int launch_cfg16(const MlpArgs& a, hipStream_t stream) {
    using RG = RingBf16;
    auto kern = mlp_fwd16_kernel<ACT, HD, MODE>;
    if (MODE == M16_STATIC_DENSITY) NERFX_REQUIRE(a.n_pieces * PIECE_UNITS == (uint32_t)m16_trunk_units(HD), "stream length %u", a.n_pieces * PIECE_UNITS);
    if (MODE == M16_STATIC_COLOUR)
        NERFX_REQUIRE(a.n_pieces * PIECE_UNITS == (uint32_t)(m16_total_units(HD) + M16_STATIC_PAD), "stream length %u", a.n_pieces * PIECE_UNITS);
    const size_t lds_bytes = (size_t)a.const_floats * sizeof(float) + (size_t)M16_RING_UNITS * UNIT_BYTES +
                             (size_t)RG::WAVES * M16_STASH_BYTES_PER_WAVE;
    NERFX_REQUIRE(lds_bytes <= 160 * 1024, "model needs %zu bytes of LDS per workgroup", lds_bytes);
    NERFX_REQUIRE(a.n_pieces * PIECE_UNITS >= 2u * M16_RING_UNITS, "stream shorter than two ring lengths");
    NERFX_OPT_IN_LDS(kern, lds_bytes);
    int dev = 0, cus = 256;
    NERFX_HIP(hipGetDevice(&dev));
    NERFX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const uint32_t grid = std::min<uint32_t>(a.n_blocks, (uint32_t)cus);      // persistent workgroups, one per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * RG::WAVES), lds_bytes, stream, a);
    return launch_status("mlp_fwd16_kernel");
}

}  // namespace nerfx
