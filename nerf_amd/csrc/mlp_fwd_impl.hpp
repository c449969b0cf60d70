// K2: the fused radiance-field MLP (encode -> trunk -> density head -> colour head).
//
// Orientation.  Every layer is computed TRANSPOSED, H^T = W . X^T, with the
// weights as the MFMA A operand (32 output features x k) and the samples as the
// 32 columns of the B operand.  A 32x32 accumulator tile then has its SAMPLE on
// the lane and its 16 FEATURES in registers, which is exactly what the next
// layer's B operand needs: activations never leave the register file between
// layers (v_mfma_f32_32x32x16_bf16 takes pairs of converted accumulator
// registers, v_mfma_f32_32x32x2_f32 takes the accumulator registers as they
// stand).  The price is a k order that is permuted per lane half; the weight
// packer (model.hip) bakes that permutation into the unit stream.
//
// Work split.  A wavefront owns 32 samples end to end; a workgroup is WAVES
// wavefronts that share the weights through LDS.  Weights arrive as a flat unit
// stream (mlp.hpp) through an LDS ring of 8 KiB pieces filled by global_load_lds,
// RING_AHEAD pieces ahead of the consumer.  A segment of k-units is handed over
// with one counted s_waitcnt + one s_barrier; workgroups are persistent and the
// stream wraps from one block of 32*WAVES samples into the next, so the ring never
// drains.
//
// The encoders are fused as producers of B operands: the sin/cos k-units of a
// sample are computed once per block into registers and consumed by layer 0, the
// skip layer and rgb.0; they are never stored.
#pragma once
#include <type_traits>

#include "mlp.hpp"

namespace nerfx {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x8 = __attribute__((ext_vector_type(8))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using lds_bytes_t = __attribute__((address_space(3))) char*;          // explicit LDS pointers: the
using lds_floats_t = const __attribute__((address_space(3))) float*;  // ring must never go through flat loads
#define NERFX_LDS_VEC(T, p) (*reinterpret_cast<const __attribute__((address_space(3))) T*>(p))

struct MlpArgs {
    const char* stream;          // packed units
    const float* consts;         // const block (global)
    const uint32_t* nan_flag;    // which outputs a NaN parameter reaches (nerfx_model::nan_flag), or NULL
    const float* ro;             // ray mode: (B,3); point mode: x (M,3)
    const float* rd;             // ray mode: (B,3); point mode: d (M,3)
    const float* t;              // ray mode: (B,n); point mode: NULL
    float* sigma;                // (M)
    float* rgb;                  // (M,3) or NULL (density only)
    uint32_t n_points;           // M
    uint32_t n_samples;          // samples per ray (ray mode)
    uint32_t n_blocks;
    uint32_t n_pieces;           // stream length for this launch, in pieces
    int hd, resid;
    int ex_kind, ex_n, ex_units, ed_kind, ed_n, ed_units;
    int sigw_off, sigb_off, tabx_off, tabd_off, const_floats;
    int foldb_off;               // const block: rgb.0's bias with `features` folded in
    unsigned long long* stamps;  // diagnostic builds: 6 cycle buckets, else NULL
    void* tape;                  // training forward: the tape to record (see Tape), else NULL
    void* tape_signs;            // bf16 training forward: the sign words of the tape (see Tape::signs)
    uint64_t tape_m_pad;         // samples per tape tile
    int tape_ex_tile, tape_ed_tile, tape_z_tile[MAX_LAYERS];   // first tile of each encoder / layer
    int tape_tiles;              // tiles in the tape (checked builds trap on anything beyond)
};

// The const block into LDS.  When the refresh that packed the weights met a NaN parameter, the bias of
// every head that parameter reaches in the reference becomes NaN here, and with it that head's output
// for every sample: the density head's bias and the three biases of the last colour layer.
template <int THREADS>
__device__ __forceinline__ void load_const_block(const MlpArgs& a, __attribute__((address_space(3))) float* cst, int tid) {
    const uint32_t poison = a.nan_flag ? __builtin_nontemporal_load(a.nan_flag) : 0u;
    const int rgb_bias = (2 * a.hd + 2) * BIAS_STRIDE;
    const float nan = __builtin_bit_cast(float, 0x7FC00000u);
    for (int i = tid; i < a.const_floats; i += THREADS) {
        float v = a.consts[i];
        if ((poison & 1u) && i == a.sigb_off) v = nan;
        if ((poison & 2u) && i >= rgb_bias && i < rgb_bias + 3) v = nan;
        cst[i] = v;
    }
}

// ---- optional in-kernel cycle accounting (diagnostic builds only: -DNERFX_STAMPS) -------
// Bucket b accumulates the s_memtime cycles a wave spends between mark(b') and mark(b).
struct Stamps {
#ifdef NERFX_STAMPS
    unsigned long long last, sum[8];
    __device__ __forceinline__ void start() { last = __builtin_amdgcn_s_memtime(); for (int i = 0; i < 8; ++i) sum[i] = 0; }
    __device__ __forceinline__ void fine(int b) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        sum[b] += now - last; last = now;
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void mark(int b) {
#if NERFX_STAMPS != 2
        fine(b);
#else
        (void)b;
#endif
    }
    // NERFX_STAMPS=2: only the marks made with coarse() count (a handful per block: the clock reads
    // of the fine marks cost about a tenth of the kernel's time and move what they measure).
    __device__ __forceinline__ void coarse(int b) {
#if NERFX_STAMPS == 2
        fine(b);
#else
        (void)b;
#endif
    }
    __device__ __forceinline__ void flush(unsigned long long* out, int lane) {
        if (lane == 0 && out) for (int i = 0; i < 8; ++i) atomicAdd(out + i, sum[i]);
    }
    // (kernels that cut their run blocks at the barriers book the waits there: kind 0 = the barrier that
    // opens this wave's run, 1 = the one that closes it)
    int phase = 0;
    __device__ __forceinline__ void mark_wait(int bucket, int) { mark(bucket); }
#else
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void coarse(int) {}
    __device__ __forceinline__ void flush(unsigned long long*, int) {}
    __device__ __forceinline__ void mark_wait(int, int) {}
    int phase = 0;
#endif
};
// (ST_BARRIER: the closing barrier of a run, booked apart by the kernels whose blocks can be cut there)
enum { ST_PROLOGUE = 0, ST_ACQUIRE = 1, ST_RUN = 2, ST_EPILOGUE = 3, ST_HEADS = 4, ST_ENCODE = 5, ST_BARRIER = 6 };

// ---- training tape -------------------------------------------------------------------------
// Scratch in HBM shared by the training forward, the backward chain and the weight-gradient
// kernel: an array of TILES [tile][sample][32 positions] (bf16 or fp32 with the kernel's
// precision; `m_pad` samples per tile).  A tile is 32 features of every sample, stored the way
// the producing wavefront holds it: lane half h of a sample's lane owns positions 16 h .. 16 h + 15
// = its 16 accumulator registers in order, so register r of half h (feature
// (r & 3) + 8 (r >> 2) + 4 h of the tile) sits at position 16 h + r, a lane moves 32 (64) contiguous
// bytes and a wavefront 2 (4) KiB.  Encoder tiles hold 32 consecutive encoder slots in slot order.
#ifndef NERFX_TAPE_NT_LEVEL
#define NERFX_TAPE_NT_LEVEL 1
#endif
// Tape stores of class LEVEL (1: tiles and encoder slots, 2: sign words) are non-temporal up to
// NERFX_TAPE_NT_LEVEL.  The tiles (3 GB per 4,096-ray step) are far larger than the caches and are
// next read by another kernel: stored with the default policy the backward chain took 408 instead
// of 365 us per step and the weight gradient 709 instead of 685.  The sign words (0.1 GB) are read
// back by the backward chain while they are still cached: non-temporal, the chain is back at 413.
template <int LEVEL, class V>
__device__ __forceinline__ void tape_store(const V& v, V* p) {
    if constexpr (LEVEL <= NERFX_TAPE_NT_LEVEL) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int PREC>
struct Tape {
    static constexpr bool BF16 = PREC == NERFX_PREC_BF16;
    using T = typename std::conditional<BF16, __bf16, float>::type;
    using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
    T* base;
    uint64_t m_pad;
    uint32_t m;                  // this lane's sample
    // bf16 training with ReLU-like activations: beside every z tile, one 16-bit word per sample
    // and lane half holding the signs of its 16 values (bit 15 - r = sign bit of register r).  The
    // backward chain needs nothing else of z, so it reads 2 bytes where the tile has 32.
    uint16_t* signs = nullptr;
    uint32_t n_tiles = 0;        // tiles the tape holds (0: not known, nothing is checked)
    // Every tape address goes through here.  A checked build (`make checked`, -DNERFX_CHECK_TAPE) traps
    // on a tile or sample outside the tape instead of letting the access fault the GPU or land in a
    // neighbouring allocation; the product build compiles this to nothing.
    __device__ __forceinline__ void check(int tile, int pos) const {
#ifdef NERFX_CHECK_TAPE
        if (n_tiles && ((uint32_t)tile >= n_tiles || (uint64_t)m >= m_pad || (uint32_t)pos >= 32u)) {
            printf("nerfx: tape access outside the tape: tile %d of %u, sample %u of %llu, position %d (block %u, thread %u)\n",
                   tile, n_tiles, m, (unsigned long long)m_pad, pos, blockIdx.x, threadIdx.x);
            __builtin_trap();
        }
#else
        (void)tile; (void)pos;
#endif
    }
    __device__ __forceinline__ uint16_t* sign_at(int tile, int h) const { check(tile, h); return signs + ((uint64_t)tile * m_pad + m) * 2 + h; }
    static __device__ __forceinline__ uint32_t sign_word(const f32x16& v) {
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = v[r];          // (bit_cast of a vector-element lvalue reads element 0 with this hipcc)
            bits = __builtin_amdgcn_alignbit(bits, __builtin_bit_cast(uint32_t, e), 31);
        }
        return bits & 0xFFFFu;
    }
    __device__ __forceinline__ T* at(int tile, int pos) const { check(tile, pos); return base + ((uint64_t)tile * m_pad + m) * 32 + pos; }
    // (callers make sure `v` does not come straight out of an MFMA: the bf16 form converts in asm)
    __device__ __forceinline__ void store_tile(int tile, int h, const f32x16& v) const {
        T* p = at(tile, 16 * h);
        if constexpr (BF16) {
            auto pk = [](float lo, float hi) { uint32_t r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; };
            u32x4 p0 = u32x4{pk(v[0], v[1]), pk(v[2], v[3]), pk(v[4], v[5]), pk(v[6], v[7])};
            u32x4 p1 = u32x4{pk(v[8], v[9]), pk(v[10], v[11]), pk(v[12], v[13]), pk(v[14], v[15])};
#ifdef NERFX_TAPE_STORE_SPLIT
            // a lane's 32 bytes as its own two 16-byte stores: each instruction fills every other
            // 16 bytes of the wavefront's 2 KiB
            reinterpret_cast<u32x4*>(p)[0] = p0;
            reinterpret_cast<u32x4*>(p)[1] = p1;
#else
            // The two halves trade places between the 16-lane rows first (row 1 of the first half
            // with row 0 of the second, row 3 with row 2), so that the first store carries all 64
            // bytes of the wavefront's samples 0..15 and the second those of samples 16..31: two
            // contiguous 1 KiB stores.  Row r of a store then holds, for sample 16 * store + lane % 16,
            // the piece (lane half r / 2, 16-byte half r % 2) -- the same bytes at the same addresses.
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(p0[e], p1[e], false, false);
                p0[e] = sw[0]; p1[e] = sw[1];
            }
            const int k = (int)((threadIdx.x >> 4) & 1);                // this lane's row is an odd one
            T* q = p - 16 * k * 32 + 8 * k;                             // sample m - 16 k, position 16 h + 8 k
            // (non-temporal: the tape is far larger than the caches and is next read by another kernel;
            // with the default policy the backward chain took 413 instead of 373 us per step, the weight
            // gradient 715 instead of 692)
            tape_store<1>(p0, reinterpret_cast<u32x4*>(q));
            tape_store<1>(p1, reinterpret_cast<u32x4*>(q + 16 * 32));
#endif
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) tape_store<1>(f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}, reinterpret_cast<f32x4*>(p) + q);
        }
    }
    // 16-byte accesses per lane and tile: the fewest instructions a tile's load or store can take
    static constexpr int TILE_VMEM_OPS = BF16 ? 2 : 4;
    // A bf16 tile as loaded, for loads that are issued long before their values are needed.
    // hipcc drains the whole vector-memory counter at the first use of an ordinary load's result
    // while LDS-DMA transfers are in flight, so these two loads are issued from asm, which the
    // compiler does not track: the CALLER waits (s_waitcnt vmcnt(n), n = vector-memory
    // instructions it has issued since) before it hands the tile to expand().
    struct Raw { u32x4 w[2]; };
    __device__ __forceinline__ Raw load_raw_async(int tile, int h) const {
        static_assert(BF16, "raw tiles are the bf16 form");
        const T* p = at(tile, 16 * h);
        Raw r;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                     : "=&v"(r.w[0]), "=&v"(r.w[1]) : "v"(p) : "memory");
        return r;
    }
    // The values of a raw tile.  The empty asm pins the registers: nothing derived from them is
    // computed before this point.
    static __device__ __forceinline__ f32x16 expand(Raw r) {
        asm volatile("" : "+v"(r.w[0]), "+v"(r.w[1]));
        f32x16 v;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[8 * q + 2 * e] = __builtin_bit_cast(float, r.w[q][e] << 16);
                v[8 * q + 2 * e + 1] = __builtin_bit_cast(float, r.w[q][e] & 0xFFFF0000u);
            }
        return v;
    }
    __device__ __forceinline__ f32x16 load_tile(int tile, int h) const {
        const T* p = at(tile, 16 * h);
        f32x16 v;
        if constexpr (BF16) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32x4 w = reinterpret_cast<const u32x4*>(p)[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[8 * q + 2 * e] = __builtin_bit_cast(float, w[e] << 16);
                    v[8 * q + 2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xFFFF0000u);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = reinterpret_cast<const f32x4*>(p)[q];
                v[4 * q] = w[0]; v[4 * q + 1] = w[1]; v[4 * q + 2] = w[2]; v[4 * q + 3] = w[3];
            }
        }
        return v;
    }
    // B operand of encoder k-unit u (16 slots in bf16, 8 in fp32; lane half h supplies half of them)
    template <class Frag>
    __device__ __forceinline__ void store_enc(int tile0, int u, int h, const Frag& f) const {
        constexpr int SPU = BF16 ? 16 : 8;
        const int slot = SPU * u + (SPU / 2) * h;
        tape_store<1>(f, reinterpret_cast<Frag*>(at(tile0 + slot / 32, slot % 32)));
    }
};

// ---- LDS weight ring ---------------------------------------------------------------
template <class RG>
struct WeightRing {
    static constexpr int WAVES = RG::WAVES, RING_PIECES = RG::PIECES, RING_AHEAD = RG::AHEAD, SYNC_PIECES = RG::SYNC;
    static constexpr int LPW = PIECE_UNITS / WAVES;         // global_load_lds per wave per piece
    static_assert(PIECE_UNITS % WAVES == 0, "a piece must split evenly over the waves");
    const char* src;             // stream base (wave-uniform; lanes add their own 32-bit offset)
    lds_bytes_t lds;             // ring base
    uint32_t n_pieces, issue_piece, issue_slot;
    uint32_t piece;              // consumer position: ring slot of the current segment's first piece
    int wave, lane;

    __device__ __forceinline__ void issue() {
#pragma unroll
        for (int k = 0; k < LPW; ++k) {
            // uniform 64-bit base + 32-bit lane offset: the lane part is cheap to rebuild, so nothing
            // 64-bit and per-lane has to stay live between hand-overs
            const char* g = src + ((size_t)issue_piece * PIECE_UNITS + wave * LPW + k) * UNIT_BYTES + (uint32_t)(lane * 16);
            lds_bytes_t l = lds + (issue_slot * PIECE_UNITS + wave * LPW + k) * UNIT_BYTES;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                             (void __attribute__((address_space(3)))*)l, 16, 0, 0);
        }
        issue_piece = issue_piece + 1 == n_pieces ? 0 : issue_piece + 1;
        issue_slot = wrap(issue_slot + 1);
    }
    // slot index modulo the ring (a mask when the ring has a power-of-two number of pieces)
    static __device__ __forceinline__ uint32_t wrap(uint32_t slot) {
        if constexpr ((RING_PIECES & (RING_PIECES - 1)) == 0) return slot & (RING_PIECES - 1);
        else return slot >= RING_PIECES ? slot - RING_PIECES : slot;
    }
    __device__ __forceinline__ void init(const char* s, lds_bytes_t l, uint32_t np, int w, int ln) {
        src = s;
        lds = l; n_pieces = np; wave = w; lane = ln;
        issue_piece = 0; issue_slot = 0; piece = 0;
#pragma unroll
        for (int c = 0; c < RING_AHEAD; ++c) issue();
    }
    // Hand-over of the next N pieces: wait for this wave's share of them (everything but the
    // (RING_AHEAD - N) newest pieces has landed), meet the other waves -- their shares have
    // landed too and nobody reads anything older any more -- then refill the N slots that
    // fell out of the window.
    // YOUNGER: vector-memory instructions (tape loads and stores) this wave has issued since its
    // last issue() and which may stay in flight across the hand-over.  The counter retires in issue
    // order, so the newest (RING_AHEAD - N) pieces plus those YOUNGER instructions may remain.  It
    // must not exceed the true number (a lower bound is safe: the wait only gets stricter).
    template <int N, int YOUNGER = 0>
    __device__ __forceinline__ void acquire() {
        static_assert(N >= 1 && N <= SYNC_PIECES, "hand-over larger than the ring's slack");
        static_assert(YOUNGER >= 0 && (RING_AHEAD - N) * LPW + YOUNGER <= 63, "vmcnt is a 6-bit field");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING_AHEAD - N) * LPW + YOUNGER) : "memory");
        arrive<N>();
    }
    // The hand-over without its wait, for callers that pick the wait themselves (wait_for<N, YOUNGER>).
    template <int N, int YOUNGER>
    static __device__ __forceinline__ void wait_for() {
        static_assert(YOUNGER >= 0 && (RING_AHEAD - N) * LPW + YOUNGER <= 63, "vmcnt is a 6-bit field");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING_AHEAD - N) * LPW + YOUNGER) : "memory");
    }
    template <int N>
    __device__ __forceinline__ void arrive() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < N; ++i) issue();
    }
    template <int N>
    __device__ __forceinline__ void release() { piece = wrap(piece + N); }
    // Address of unit `rel` (compile-time) of the segment that starts at `piece`.
    __device__ __forceinline__ lds_bytes_t unit_ptr(int rel) const {
        const uint32_t slot = wrap(piece + rel / PIECE_UNITS);
        return lds + slot * (PIECE_UNITS * UNIT_BYTES) + (rel % PIECE_UNITS) * UNIT_BYTES + lane * 16;
    }
};

// ---- per-precision pieces ------------------------------------------------------------
template <int PREC> struct Prec;

template <> struct Prec<NERFX_PREC_BF16> {
    static constexpr int SPUH = 8;                          // elements a lane supplies per k-unit
    static constexpr int UPT = 2;
    using Frag = bf16x8;                                    // one B operand per k-unit
    // Two floats -> one dword of two bf16 (round to nearest even, NaN stays NaN).  Spelled out
    // because hipcc converts element by element (one v_cvt_pk per float plus a v_perm per pair).
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        uint32_t r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
        return r;
    }
    // (the inputs of pack2 must not come straight from an MFMA or a transcendental instruction:
    // hipcc pads no hazard for an asm consumer -- from_floats therefore converts the plain way)
    static __device__ __forceinline__ Frag from_floats(const float (&v)[8]) {
        Frag f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (__bf16)v[e];
        return f;
    }
    // accumulator registers 8s..8s+7 of tile `a` -> B operand of chained k-unit 2T+s
    static __device__ __forceinline__ Frag from_acc(const f32x16& a, int s) {
        using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
        const u32x4 p = {pack2(a[8 * s], a[8 * s + 1]), pack2(a[8 * s + 2], a[8 * s + 3]),
                         pack2(a[8 * s + 4], a[8 * s + 5]), pack2(a[8 * s + 6], a[8 * s + 7])};
        return __builtin_bit_cast(Frag, p);
    }
    // ReLU on packed bf16: a negative float is a negative int16 (v_pk_max_i16 with 0)
    static constexpr bool PACKED_RELU = true;
    static __device__ __forceinline__ Frag relu(const Frag& f) {
        const s16x8 v = __builtin_bit_cast(s16x8, f);
        return __builtin_bit_cast(Frag, __builtin_elementwise_max(v, s16x8(0)));
    }
    using AFrag = bf16x8;                                   // weights of one unit for this lane
    static constexpr int MFMA_PER_UNIT = 1;
    static __device__ __forceinline__ AFrag load_a(lds_bytes_t unit) { return NERFX_LDS_VEC(bf16x8, unit); }
    static __device__ __forceinline__ void mma(f32x16& acc, const AFrag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    // The hardware sine / cosine take revolutions: the coordinates are scaled once per sample
    // (ANGLE_UNIT) so that the pair table's dot product comes out in revolutions.
    static constexpr float ANGLE_UNIT = 0.15915494309189535f;
    static __device__ __forceinline__ void sincos_pair(float revolutions, float& s, float& c) {
        const float r = __builtin_amdgcn_fractf(revolutions);
        s = __builtin_amdgcn_sinf(r);
        c = __builtin_amdgcn_cosf(r);
    }
    static __device__ __forceinline__ float expneg(float x) { return __expf(-x); }
    // x * sigmoid(x) with the hardware exp and reciprocal (1 ulp each, far inside bf16's 8 bits):
    // five instructions per value where an IEEE division alone costs ten
    static __device__ __forceinline__ float sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
    static __device__ __forceinline__ float silu(float x) { return x * sigmoid(x); }
    // output activations with hardware exp / log / rcp: their ~1e-6 error is far inside bf16's
    // (log(1 + e) rounds to 0 once e < 2^-24; log1p(e) = e to fp32 precision below 2.4e-4)
    static __device__ __forceinline__ float softplus(float x) {
        const float e = __expf(-fabsf(x));
        return (e < 2.4e-4f ? e : __logf(1.0f + e)) + fmaxf(x, 0.0f);
    }
    static __device__ __forceinline__ float widened_sigmoid(float x) {
        return fmaf(1.002f, __frcp_rn(1.0f + __expf(-x)), -0.001f);
    }
};

template <> struct Prec<NERFX_PREC_FP32> {
    static constexpr int SPUH = 4;
    static constexpr int UPT = 4;
    using Frag = f32x4;                                     // four B operands (k-steps of 2) per k-unit
    static __device__ __forceinline__ Frag from_floats(const float (&v)[4]) { return Frag{v[0], v[1], v[2], v[3]}; }
    // accumulator registers 4g..4g+3 of tile `a` -> B operands of chained k-unit 4T+g
    static __device__ __forceinline__ Frag from_acc(const f32x16& a, int g) {
        return Frag{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
    }
    static constexpr bool PACKED_RELU = false;
    static __device__ __forceinline__ Frag relu(const Frag& f) { return f; }
    using AFrag = f32x4;
    static constexpr int MFMA_PER_UNIT = 4;
    static __device__ __forceinline__ AFrag load_a(lds_bytes_t unit) { return NERFX_LDS_VEC(f32x4, unit); }
    static __device__ __forceinline__ void mma(f32x16& acc, const AFrag& a, const Frag& b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    }
    static constexpr float ANGLE_UNIT = 1.0f;              // radians
    static __device__ __forceinline__ void sincos_pair(float angle, float& s, float& c) { sincosf(angle, &s, &c); }
    static __device__ __forceinline__ float expneg(float x) { return expf(-x); }
    static __device__ __forceinline__ float sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
    static __device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
    // activations.py:33-35 and :18-20, evaluated as the reference writes them
    static __device__ __forceinline__ float softplus(float x) { return log1pf(expf(-fabsf(x))) + (x >= 0.0f ? x : 0.0f); }
    static __device__ __forceinline__ float widened_sigmoid(float x) { return 0.5f * (1.0f + 1.002f * tanhf(0.5f * x)); }
};

// B operand of encoder k-unit u for this lane's point: slots per enc_unit_slot() / enc_slot_col().
// `table` is the encoder's pair-coefficient table (float4 per pair-slot, LDS); (sx, sy, sz) is
// the point as the angles want it (v for positional, 2 pi v for Fourier: features.py:34, :51;
// times the precision's ANGLE_UNIT); a positional encoder's unit 0 starts with the raw
// coordinates (x, y, z) in lane half 0.
template <int PREC>
__device__ __forceinline__ typename Prec<PREC>::Frag encode_unit(lds_floats_t table, bool positional, int u, int h,
                                                                 float sx, float sy, float sz, float x, float y, float z) {
    using P = Prec<PREC>;
    constexpr int SPUH = P::SPUH;
    const lds_floats_t c0 = table + 4 * ((SPUH * u) + (SPUH / 2) * h);       // this lane's first pair-slot
    float v[SPUH];
#pragma unroll
    for (int i = 0; i < SPUH / 2; ++i) {
        const f32x4 c = NERFX_LDS_VEC(f32x4, c0 + 4 * i);
        const float angle = fmaf(sz, c[2], fmaf(sy, c[1], __fmul_rn(sx, c[0])));
        P::sincos_pair(angle, v[2 * i], v[2 * i + 1]);
    }
    if (positional && u == 0 && h == 0) { v[0] = x; v[1] = y; v[2] = z; v[3] = 0.0f; }
    return P::from_floats(v);
}

// An encoder's k-units for one sample: the first ENC_BLOCK live in registers (that covers
// positional encodings up to 10 frequencies); wider encoders are re-evaluated on use.
template <int PREC>
struct Encoding {
    typename Prec<PREC>::Frag cached[ENC_BLOCK];
    int units;
    bool positional;
    lds_floats_t table;
    float sx, sy, sz, rx, ry, rz;
    int h;
    template <bool BIGENC>
    __device__ __forceinline__ void prepare(int kind, int nu, lds_floats_t tb, int hh, float x, float y, float z) {
        positional = kind == NERFX_ENC_POSITIONAL;
        // 2 pi x ANGLE_UNIT is exactly 1 for the revolutions of the bf16 kernels
        const float scale = positional ? Prec<PREC>::ANGLE_UNIT : (PREC == NERFX_PREC_BF16 ? 1.0f : 6.283185307179586f);
        units = nu; table = tb; h = hh;
        sx = __fmul_rn(scale, x); sy = __fmul_rn(scale, y); sz = __fmul_rn(scale, z);
        rx = x; ry = y; rz = z;
        if (!BIGENC) {
#pragma unroll
            for (int u = 0; u < ENC_BLOCK; ++u) cached[u] = encode_unit<PREC>(table, positional, u, h, sx, sy, sz, rx, ry, rz);
        }
    }
    __device__ __forceinline__ typename Prec<PREC>::Frag eval(int u) const {
        return encode_unit<PREC>(table, positional, u, h, sx, sy, sz, rx, ry, rz);
    }
};

template <int PREC, int ACT>
__device__ __forceinline__ float activate(float x) {
    // max(x, 0) on the bits, as signed integers: negative floats (and -0) are negative integers.
    // fmaxf costs a second v_max_f32 wherever the compiler cannot prove its input is no signalling
    // NaN (every value that comes out of an asm block); a NaN with a clear sign bit stays a NaN here,
    // as it does in the reference (model.py:68-74, torch's relu)
    if (ACT == NERFX_ACT_RELU) return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0));
    if (ACT == NERFX_ACT_LEAKY_RELU) return fmaxf(x, 0.01f * x);       // slope < 1: the larger of the two lines
    if (ACT == NERFX_ACT_SILU) return Prec<PREC>::silu(x);
    return x;                                                // ACT < 0: plain Linear
}

#ifdef NERFX_DIAG_MFMA16
#include "mlp_runs_bf16_diag16.inc"     // timing diagnostic (wrong results): see tools/gen_mlp_runs_diag16.py
#else
#include "mlp_runs_bf16.inc"
#endif

// One acquired run of N units: unit i multiplies B operand b(i) into acc[i % NG].  The A
// operands are read LOOKAHEAD units ahead of the MFMA that consumes them; the interleave is
// pinned with sched_group_barrier (ds_read = 0x100, MFMA = 0x8) because hipcc otherwise
// serialises read -> wait -> MFMA on one register set when registers are tight.
constexpr int LOOKAHEAD = 4;
template <int PREC, class RG, int NG, int N, class BSource>
__device__ __forceinline__ void run_units(const WeightRing<RG>& ring, f32x16 (&acc)[NG], BSource b) {
    using P = Prec<PREC>;
    if constexpr (PREC == NERFX_PREC_BF16 && NG == 2 && (N == 8 || N == 16 || N == 32)) {
        // hand-scheduled block: counted lgkmcnt waits (see mlp_runs_bf16.inc)
        uint32_t p[N / PIECE_UNITS];
#pragma unroll
        for (int k = 0; k < N / PIECE_UNITS; ++k) p[k] = (uint32_t)(uintptr_t)ring.unit_ptr(k * PIECE_UNITS);
        // the wave inside a hand-scheduled run goes first on its SIMD: the other wave's epilogue fills
        // the slots between the MFMAs instead of delaying them (render: -0.5 %; priority 2 or 3 alike)
        __builtin_amdgcn_s_setprio(3);
        if constexpr (N == 8) run8_bf16_ng2(acc[0], acc[1], p[0], b(0), b(1), b(2), b(3));
        if constexpr (N == 16) run16_bf16_ng2(acc[0], acc[1], p[0], p[1], b(0), b(1), b(2), b(3), b(4), b(5), b(6), b(7));
        if constexpr (N == 32)
            run32_bf16_ng2(acc[0], acc[1], p[0], p[1], p[2], p[3], b(0), b(1), b(2), b(3), b(4), b(5), b(6), b(7),
                           b(8), b(9), b(10), b(11), b(12), b(13), b(14), b(15));
        __builtin_amdgcn_s_setprio(0);
        return;
    }
    __builtin_amdgcn_sched_barrier(0);
    typename P::AFrag a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = P::load_a(ring.unit_ptr(i));
#pragma unroll
    for (int i = 0; i < N; ++i) P::mma(acc[i % NG], a[i], b(i / NG));
    constexpr int D = LOOKAHEAD < N ? LOOKAHEAD : N;
    __builtin_amdgcn_sched_group_barrier(0x100, D, 0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, P::MFMA_PER_UNIT, 0);
        if (i + D < N) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PREC == NERFX_PREC_BF16) {
        // the epilogue's asm conversions (pack2) read these accumulators: let the last MFMA finish,
        // as the hand-scheduled blocks do with their closing nops
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]));
#pragma unroll
        for (int t = 1; t < NG; ++t) asm volatile("" : "+v"(acc[t]));
    }
}

// Whether a kernel configuration keeps the positional encoding of x in the LDS stash.
template <int PREC, int WT, bool BIGENC>
constexpr bool uses_stash() { return PREC == NERFX_PREC_BF16 && !BIGENC && WT >= 2; }

// ---- one Linear layer -------------------------------------------------------------------
// out = act(W . [encoder units | chained input] + b), computed NG output tiles at a time so
// that only NG accumulator tiles are live beside the input and output fragments.
//   hin  : B operands of the chained input (CT * UPT fragments used)
//   hout : receives the activated output as next-layer B operands (NT * UPT fragments)
//   with_enc: this layer's stream carries an encoder segment (wave-uniform)
//   SIGMA: also accumulate the density head's dot product on the fp32 activations
//   raw  : if non-null (NT == 1 only) receives the fp32 accumulator tile instead of hout
//   ztape: training forward (TRAIN): the pre-activation output goes to tape tiles ztile..;
//          with an encoder segment, its B operands go to tiles etile.. in slot order
//   PREV_OPS (TRAIN): tape stores the caller's previous step has left in flight (a lower bound);
//          the first hand-over of a group lets them, or the previous group's, stay in flight
template <int PREC, class RG, int ACT, bool BIGENC, int NT, int CT, bool SIGMA, bool STASH = false, bool TRAIN = false,
          int PREV_OPS = 0, int HN_IN = 0, int HN_OUT = 0>
__device__ __forceinline__ float linear(WeightRing<RG>& ring, lds_floats_t bias, int h,
                                        const typename Prec<PREC>::Frag (&hin)[HN_IN],
                                        typename Prec<PREC>::Frag (&hout)[HN_OUT],
                                        bool with_enc, const Encoding<PREC>& enc, lds_floats_t sigw,
                                        Stamps& st, f32x16* raw = nullptr, uint32_t stash = 0,
                                        const Tape<PREC>* ztape = nullptr, int ztile = 0, int etile = -1) {
    using P = Prec<PREC>;
    constexpr int NG = group_tiles(PREC, NT);
    constexpr int SYNC_PIECES = RG::SYNC;
    constexpr int KB_PER_PIECE = PIECE_UNITS / NG;              // k-blocks (NG units each) per piece
    constexpr int CHAIN_KU = CT * P::UPT;
    constexpr int CHAIN_PIECES = pad_to_piece(CHAIN_KU * NG) / PIECE_UNITS;
    constexpr int ENC_PIECES = pad_to_piece(ENC_BLOCK * NG) / PIECE_UNITS;   // per ENC_BLOCK k-units
    static_assert(CHAIN_KU <= HN_IN && NT * P::UPT <= HN_OUT, "fragment arrays too small");
    constexpr int STORED = TRAIN ? NG * Tape<PREC>::TILE_VMEM_OPS : 0;   // tape stores of one group, at least
    float dot = 0.0f;
#pragma unroll
    for (int grp = 0; grp < NT / NG; ++grp) {
        f32x16 acc[NG];
        // bias as the initial accumulator: register r of a tile is feature (r&3) + 8 (r>>2) + 4 h
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            lds_floats_t bp = bias + 32 * (grp * NG + t) + 4 * h;
            const f32x4 b0 = NERFX_LDS_VEC(f32x4, bp), b1 = NERFX_LDS_VEC(f32x4, bp + 8),
                        b2 = NERFX_LDS_VEC(f32x4, bp + 16), b3 = NERFX_LDS_VEC(f32x4, bp + 24);
            const f32x8 lo = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            const f32x8 hi = __builtin_shufflevector(b2, b3, 0, 1, 2, 3, 4, 5, 6, 7);
            acc[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        }
        if (with_enc) {
            if (!BIGENC) {                                         // register-resident encoding
                st.mark(ST_EPILOGUE);
                if constexpr (TRAIN) {
                    if (grp == 0) ring.template wait_for<ENC_PIECES, PREV_OPS>();
                    else ring.template wait_for<ENC_PIECES, STORED>();
                    ring.template arrive<ENC_PIECES>();
                } else {
                    ring.template acquire<ENC_PIECES>();
                }
                st.mark(ST_ACQUIRE);
                if constexpr (STASH) {                             // B operands from the wave's LDS stash
                    static_assert(PREC == NERFX_PREC_BF16 && NG == 2, "stash blocks exist for bf16, two tiles");
                    // `stash` is the wave-uniform byte offset of this wave's stash; the lane part is
                    // re-added here so that no per-lane address has to stay live across layers
                    run8_bf16_ng2_stash(acc[0], acc[1], (uint32_t)(uintptr_t)ring.unit_ptr(0), stash,
                                        (uint32_t)ring.lane * 16u);
                } else {
                    run_units<PREC, RG, NG, ENC_BLOCK * NG>(ring, acc, [&](int u) { return enc.cached[u]; });
                }
                st.mark(ST_RUN);
                ring.template release<ENC_PIECES>();
            } else {
                for (int u0 = 0; u0 < enc.units; u0 += ENC_BLOCK) {
                    typename P::Frag eb[ENC_BLOCK];
#pragma unroll
                    for (int u = 0; u < ENC_BLOCK; ++u) eb[u] = enc.eval(u0 + u);
                    if constexpr (TRAIN) {
                        if (grp == 0 && etile >= 0) {
#pragma unroll
                            for (int u = 0; u < ENC_BLOCK; ++u) ztape->store_enc(etile, u0 + u, h, eb[u]);
                        }
                    }
                    ring.template acquire<ENC_PIECES>();
                    run_units<PREC, RG, NG, ENC_BLOCK * NG>(ring, acc, [&](int u) { return eb[u]; });
                    ring.template release<ENC_PIECES>();
                }
            }
        }
        // chained input, handed over SYNC_PIECES pieces at a time
        if constexpr (CHAIN_KU > 0)
#pragma unroll
        for (int p0 = 0; p0 < CHAIN_PIECES; p0 += SYNC_PIECES) {
            constexpr int TAIL = CHAIN_PIECES % SYNC_PIECES ? CHAIN_PIECES % SYNC_PIECES : SYNC_PIECES;
            const int ku0 = p0 * KB_PER_PIECE;
            if (CHAIN_PIECES - p0 >= SYNC_PIECES) {
                constexpr int KU = SYNC_PIECES * KB_PER_PIECE;
                st.mark(ST_EPILOGUE);
                if constexpr (TRAIN) {                             // the group's first hand-over lets tape stores fly
                    if (p0 == 0 && !with_enc) ring.template wait_for<SYNC_PIECES, (PREV_OPS < STORED ? PREV_OPS : STORED)>();
                    else ring.template wait_for<SYNC_PIECES, 0>();
                    ring.template arrive<SYNC_PIECES>();
                } else {
                    ring.template acquire<SYNC_PIECES>();
                }
                st.mark(ST_ACQUIRE);
                run_units<PREC, RG, NG, KU * NG>(ring, acc, [&](int k) { return hin[ku0 + k]; });
                st.mark(ST_RUN);
                ring.template release<SYNC_PIECES>();
            } else {
                constexpr int KU = CHAIN_KU - (CHAIN_PIECES - TAIL) * KB_PER_PIECE;   // k-units in the tail run
                st.mark(ST_EPILOGUE);
                if constexpr (TRAIN) {                             // the group's first hand-over lets tape stores fly
                    if (p0 == 0 && !with_enc) ring.template wait_for<TAIL, (PREV_OPS < STORED ? PREV_OPS : STORED)>();
                    else ring.template wait_for<TAIL, 0>();
                    ring.template arrive<TAIL>();
                } else {
                    ring.template acquire<TAIL>();
                }
                st.mark(ST_ACQUIRE);
                run_units<PREC, RG, NG, (KU > 0 ? KU : 1) * NG>(ring, acc, [&](int k) { return hin[ku0 + k]; });
                st.mark(ST_RUN);
                ring.template release<TAIL>();
            }
        }
        if constexpr (TRAIN) {
#pragma unroll
            for (int t = 0; t < NG; ++t) ztape->store_tile(ztile + grp * NG + t, h, acc[t]);
            if constexpr (tape_has_signs(PREC, ACT)) {
#pragma unroll
                for (int t = 0; t < NG; ++t)
                    tape_store<2>((uint16_t)Tape<PREC>::sign_word(acc[t]), ztape->sign_at(ztile + grp * NG + t, h));
            }
        }
        // epilogue: activation, density dot, conversion to the next layer's operands
        constexpr bool packed_relu = ACT == NERFX_ACT_RELU && P::PACKED_RELU && !SIGMA;
        if (!packed_relu) {
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = activate<PREC, ACT>(acc[t][r]);
        }
        if (SIGMA) {
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w = NERFX_LDS_VEC(f32x4, sigw + 32 * (grp * NG + t) + 8 * g + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dot = fmaf(acc[t][4 * g + i], w[i], dot);
                }
        }
        if (raw) {
            *raw = acc[0];
        } else {
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int s = 0; s < P::UPT; ++s) {
                    const typename P::Frag f = P::from_acc(acc[t], s);
                    hout[(grp * NG + t) * P::UPT + s] = packed_relu ? P::relu(f) : f;
                }
        }
    }
    return dot;
}

// A Linear layer with an encoder segment, read from the encoder-major stream
// (NERFX_STREAM_ENCMAJOR; bf16 inference): every chunk of ENC_BLOCK encoder k-units is evaluated
// ONCE and multiplied into all NT accumulator tiles of the layer -- one hand-over brings the
// chunk's weights for every tile group -- where linear<> evaluates it once per tile group.  The
// chained input then goes group by group as in linear<>, each group followed by its epilogue, so
// the NT accumulator tiles are live beside hin only while the encoder segment runs.
template <class RG, int ACT, int NT, int CT, bool SIGMA, int HN_IN, int HN_OUT>
__device__ __forceinline__ float linear_encmajor(WeightRing<RG>& ring, lds_floats_t bias, int h,
                                                 const Prec<NERFX_PREC_BF16>::Frag (&hin)[HN_IN],
                                                 Prec<NERFX_PREC_BF16>::Frag (&hout)[HN_OUT],
                                                 const Encoding<NERFX_PREC_BF16>& enc, lds_floats_t sigw, Stamps& st) {
    constexpr int PREC = NERFX_PREC_BF16;
    using P = Prec<PREC>;
    constexpr int NG = group_tiles(PREC, NT), GROUPS = NT / NG;
    constexpr int SYNC_PIECES = RG::SYNC;
    constexpr int KB_PER_PIECE = PIECE_UNITS / NG;
    constexpr int CHAIN_KU = CT * P::UPT;
    constexpr int CHAIN_PIECES = pad_to_piece(CHAIN_KU * NG) / PIECE_UNITS;
    constexpr int ENC_PIECES = pad_to_piece(ENC_BLOCK * NG) / PIECE_UNITS;   // per chunk and group
    static_assert(GROUPS * ENC_PIECES <= SYNC_PIECES, "a chunk's weights for every group arrive in one hand-over");
    static_assert(CHAIN_PIECES % SYNC_PIECES == 0, "chained segments are whole hand-overs");
    static_assert(CHAIN_KU <= HN_IN && NT * P::UPT <= HN_OUT, "fragment arrays too small");
    f32x16 acc[GROUPS][NG];
#pragma unroll
    for (int grp = 0; grp < GROUPS; ++grp)
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            lds_floats_t bp = bias + 32 * (grp * NG + t) + 4 * h;
            const f32x4 b0 = NERFX_LDS_VEC(f32x4, bp), b1 = NERFX_LDS_VEC(f32x4, bp + 8),
                        b2 = NERFX_LDS_VEC(f32x4, bp + 16), b3 = NERFX_LDS_VEC(f32x4, bp + 24);
            const f32x8 lo = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            const f32x8 hi = __builtin_shufflevector(b2, b3, 0, 1, 2, 3, 4, 5, 6, 7);
            acc[grp][t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        }
    for (int u0 = 0; u0 < enc.units; u0 += ENC_BLOCK) {
        st.mark(ST_EPILOGUE);
        typename P::Frag eb[ENC_BLOCK];
#pragma unroll
        for (int u = 0; u < ENC_BLOCK; ++u) eb[u] = enc.eval(u0 + u);
        st.mark(ST_ENCODE);
        ring.template acquire<GROUPS * ENC_PIECES>();
        st.mark(ST_ACQUIRE);
#pragma unroll
        for (int grp = 0; grp < GROUPS; ++grp) {
            run_units<PREC, RG, NG, ENC_BLOCK * NG>(ring, acc[grp], [&](int u) { return eb[u]; });
            ring.template release<ENC_PIECES>();
        }
        st.mark(ST_RUN);
    }
    float dot = 0.0f;
#pragma unroll
    for (int grp = 0; grp < GROUPS; ++grp) {
        if constexpr (CHAIN_KU > 0)
#pragma unroll
        for (int p0 = 0; p0 < CHAIN_PIECES; p0 += SYNC_PIECES) {
            const int ku0 = p0 * KB_PER_PIECE;
            constexpr int KU = SYNC_PIECES * KB_PER_PIECE;
            st.mark(ST_EPILOGUE);
            ring.template acquire<SYNC_PIECES>();
            st.mark(ST_ACQUIRE);
            run_units<PREC, RG, NG, KU * NG>(ring, acc[grp], [&](int k) { return hin[ku0 + k]; });
            st.mark(ST_RUN);
            ring.template release<SYNC_PIECES>();
        }
        constexpr bool packed_relu = ACT == NERFX_ACT_RELU && P::PACKED_RELU && !SIGMA;
        if (!packed_relu) {
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[grp][t][r] = activate<PREC, ACT>(acc[grp][t][r]);
        }
        if (SIGMA) {
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w = NERFX_LDS_VEC(f32x4, sigw + 32 * (grp * NG + t) + 8 * g + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dot = fmaf(acc[grp][t][4 * g + i], w[i], dot);
                }
        }
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
            for (int s = 0; s < P::UPT; ++s) {
                const typename P::Frag f = P::from_acc(acc[grp][t], s);
                hout[(grp * NG + t) * P::UPT + s] = packed_relu ? P::relu(f) : f;
            }
    }
    return dot;
}

// HD > 0 fixes the number of layers per trunk half at compile time (with the skip
// connection on): the block body is then straight-line code, which hipcc allocates without
// the loop-carried copies of the activation fragments that the runtime-depth form (HD = 0)
// needs and spills.
// TRAIN: also record the tape the backward kernels read (pre-activations and encodings).
template <int PREC, int WT, class RG, int ACT, bool BIGENC, int HD, bool TRAIN = false>
__global__ __launch_bounds__(64 * RG::WAVES, RG::BLOCKS_PER_CU * RG::WAVES / 4) void mlp_fwd_kernel(const MlpArgs a) {
    constexpr int WAVES = RG::WAVES;
    using P = Prec<PREC>;
    using Frag = typename P::Frag;
    constexpr int RT = WT > 1 ? WT / 2 : 1;
    constexpr int HN = WT * P::UPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) float* cst = (__attribute__((address_space(3))) float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;

    // LDS: [const block | per-wave encoding stash (bf16 kernels) | weight ring]
    constexpr bool STASH = uses_stash<PREC, WT, BIGENC>();
    const lds_bytes_t stash_base = (lds_bytes_t)smem + a.const_floats * (int)sizeof(float);
    const lds_bytes_t my_stash = stash_base + wave * STASH_BYTES_PER_WAVE + lane * 16;
    WeightRing<RG> ring;
    ring.init(a.stream, stash_base + (STASH ? WAVES * STASH_BYTES_PER_WAVE : 0), a.n_pieces, wave, lane);
    load_const_block<64 * WAVES>(a, cst, tid);
    __syncthreads();

    const lds_floats_t sigw = cst + a.sigw_off;
    const bool colour = a.rgb != nullptr;
    const int hd = HD > 0 ? HD : a.hd;
    const bool resid = HD > 0 ? true : a.resid != 0;
    const int last = 2 * hd - 1;                                   // last trunk layer feeds the density head

    Stamps st;
    st.start();
    for (uint32_t blk = blockIdx.x; blk < a.n_blocks; blk += gridDim.x) {
        // ---- this lane's sample (both lane halves hold the same one) ----------------
        const uint32_t m = blk * (32 * WAVES) + wave * 32 + (lane & 31);
        const bool valid = m < a.n_points;
        const uint32_t mc = valid ? m : a.n_points - 1;
        float px, py, pz, dx, dy, dz;
        if (a.t) {                                                  // ray.py:132 / :226
            const uint32_t ray = mc / a.n_samples;
            const float tt = a.t[mc];
            dx = a.rd[3 * ray]; dy = a.rd[3 * ray + 1]; dz = a.rd[3 * ray + 2];
            px = __fadd_rn(a.ro[3 * ray], __fmul_rn(dx, tt));
            py = __fadd_rn(a.ro[3 * ray + 1], __fmul_rn(dy, tt));
            pz = __fadd_rn(a.ro[3 * ray + 2], __fmul_rn(dz, tt));
        } else {
            px = a.ro[3 * (size_t)mc]; py = a.ro[3 * (size_t)mc + 1]; pz = a.ro[3 * (size_t)mc + 2];
            dx = a.rd[3 * (size_t)mc]; dy = a.rd[3 * (size_t)mc + 1]; dz = a.rd[3 * (size_t)mc + 2];
        }
        // A non-finite input poisons every feature in the reference (model.py:68-78).
        const float probe = (px + py + pz) + (colour ? (dx + dy + dz) : 0.0f);
        const bool poisoned = !(fabsf(probe) <= 3.0e38f);

        st.mark(ST_PROLOGUE);
        // phi_x is consumed twice (layer 0 and the skip layer).  bf16 kernels park its B operands
        // in the wave's LDS stash (the MFMA blocks read them from there); the others keep them
        // in registers, (re)built right before each consuming layer.
        Encoding<PREC> enc;
        enc.template prepare<BIGENC>(a.ex_kind, a.ex_units, cst + a.tabx_off, h, px, py, pz);
        Tape<PREC> tape{(typename Tape<PREC>::T*)a.tape, a.tape_m_pad, m, (uint16_t*)a.tape_signs, (uint32_t)a.tape_tiles};   // m < m_pad also for tail lanes
        const Tape<PREC>* zt = TRAIN ? &tape : nullptr;
        if constexpr (TRAIN && !BIGENC) {                           // wide encoders are recorded where they are evaluated
#pragma unroll
            for (int u = 0; u < ENC_BLOCK; ++u)
                if (u < a.ex_units) tape.store_enc(a.tape_ex_tile, u, h, enc.cached[u]);
        }
        uint32_t stash = 0;
        if constexpr (STASH) {
#pragma unroll
            for (int u = 0; u < ENC_BLOCK; ++u)
                *reinterpret_cast<__attribute__((address_space(3))) Frag*>(my_stash + u * UNIT_BYTES) = enc.cached[u];
            // same wave reads it back: LDS is in order per wave
            stash = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(stash_base + wave * STASH_BYTES_PER_WAVE));
        }
        st.mark(ST_ENCODE);
        Frag A[HN], B[HN];                                        // ping-pong activation fragments
        float dot = 0.0f;
        // tape stores a full-width layer's last group leaves in flight for the next layer's first hand-over
        constexpr int S_WT = TRAIN ? group_tiles(PREC, WT) * Tape<PREC>::TILE_VMEM_OPS : 0;
        // ---- fc_1 then fc_2; fc_2's first layer optionally sees [phi_x | h] (model.py:71-73).
        // Layer 0 writes A; the remaining 2 hd - 1 trunk layers alternate A -> B, B -> A, so the
        // last one always lands in B.
        // (fixed-depth bf16 inference with wide encoders reads the encoder-major stream)
        constexpr bool ENCMAJOR = BIGENC && HD > 0 && !TRAIN && PREC == NERFX_PREC_BF16;
        if constexpr (ENCMAJOR) linear_encmajor<RG, ACT, WT, 0, false>(ring, cst, h, B, A, enc, sigw, st);
        else if (last == 0) dot = linear<PREC, RG, ACT, BIGENC, WT, 0, true, STASH, TRAIN>(ring, cst, h, A, B, true, enc, sigw, st, nullptr, stash, zt, a.tape_z_tile[0], a.tape_ex_tile);
        else linear<PREC, RG, ACT, BIGENC, WT, 0, false, STASH, TRAIN>(ring, cst, h, B, A, true, enc, sigw, st, nullptr, stash, zt, a.tape_z_tile[0], a.tape_ex_tile);
        const int skip = resid ? hd : -1;                          // the layer that sees phi_x again
        auto refresh = [&](int li) {                               // register-resident encodings only
            if (!STASH && li == skip) {
                st.mark(ST_EPILOGUE);
                enc.template prepare<BIGENC>(a.ex_kind, a.ex_units, cst + a.tabx_off, h, px, py, pz);
                st.mark(ST_ENCODE);
            }
        };
        int li = 1;
#pragma unroll
        for (; li + 1 <= last; li += 2) {
            refresh(li);
            if (ENCMAJOR && li == skip) { if constexpr (ENCMAJOR) linear_encmajor<RG, ACT, WT, WT, false>(ring, cst + li * BIAS_STRIDE, h, A, B, enc, sigw, st); }
            else linear<PREC, RG, ACT, BIGENC, WT, WT, false, STASH, TRAIN, S_WT>(ring, cst + li * BIAS_STRIDE, h, A, B, li == skip, enc, sigw, st, nullptr, stash, zt, a.tape_z_tile[li]);
            refresh(li + 1);
            if (ENCMAJOR && li + 1 == skip) { if constexpr (ENCMAJOR) linear_encmajor<RG, ACT, WT, WT, false>(ring, cst + (li + 1) * BIAS_STRIDE, h, B, A, enc, sigw, st); }
            else linear<PREC, RG, ACT, BIGENC, WT, WT, false, STASH, TRAIN, S_WT>(ring, cst + (li + 1) * BIAS_STRIDE, h, B, A, li + 1 == skip, enc, sigw, st, nullptr, stash, zt, a.tape_z_tile[li + 1]);
        }
        if (last > 0) {
            refresh(last);
            dot = linear<PREC, RG, ACT, BIGENC, WT, WT, true, STASH, TRAIN, S_WT>(ring, cst + last * BIAS_STRIDE, h, A, B, last == skip, enc, sigw, st, nullptr, stash, zt, a.tape_z_tile[last]);
        }
        // ---- density head (model.py:75, activations.py:33-35) ------------------------------
        st.mark(ST_EPILOGUE);
        {
            dot += __shfl_xor(dot, 32, 64);
            const float sx = (dot + cst[a.sigb_off]) - 1.0f;
            float sg = P::softplus(sx);
            if (poisoned) sg = __builtin_nanf("");
            if (valid && h == 0) a.sigma[m] = sg;
        }
        st.mark(ST_HEADS);
        if (colour) {
            // ---- colour head.  `features` is a Linear without activation (model.py:51, :76): in the bf16
            // streams the packer has folded it into rgb.0 (LayerPlan::folded), which then reads [phi_d | h]
            // with h the last trunk layer's output (B -> A, rgb.2 reads A); fp32 evaluates the two layers
            // as the reference does (B -> A -> B, rgb.2 reads B).
            constexpr bool FOLD = folds_features(PREC);
            li = last + 1;
            if constexpr (!FOLD) linear<PREC, RG, -1, BIGENC, WT, WT, false, false, TRAIN, S_WT>(ring, cst + li * BIAS_STRIDE, h, B, A, false, enc, sigw, st, nullptr, 0, zt, a.tape_z_tile[li]);
            ++li;
            // ---- rgb.0 on [phi_d | features] (model.py:53, :76) -------------------------------
            { st.mark(ST_EPILOGUE); enc.template prepare<BIGENC>(a.ed_kind, a.ed_units, cst + a.tabd_off, h, dx, dy, dz); st.mark(ST_ENCODE); }
            if constexpr (TRAIN && !BIGENC) {
#pragma unroll
                for (int u = 0; u < ENC_BLOCK; ++u)
                    if (u < a.ed_units) tape.store_enc(a.tape_ed_tile, u, h, enc.cached[u]);
            }
            const lds_floats_t rgb0_bias = FOLD ? cst + a.foldb_off : cst + li * BIAS_STRIDE;
            Frag (&rgb0_in)[HN] = FOLD ? B : A;
            Frag (&rgb0_out)[HN] = FOLD ? A : B;
            if constexpr (ENCMAJOR) linear_encmajor<RG, ACT, RT, WT, false>(ring, rgb0_bias, h, rgb0_in, rgb0_out, enc, sigw, st);
            else linear<PREC, RG, ACT, BIGENC, RT, WT, false, false, TRAIN, S_WT>(ring, rgb0_bias, h, rgb0_in, rgb0_out, true, enc, sigw, st, nullptr, 0, zt, a.tape_z_tile[li], a.tape_ed_tile);
            ++li;
            // ---- rgb.2 + widened sigmoid (model.py:54, activations.py:18-20) ----------------
            f32x16 out;
            linear<PREC, RG, -1, BIGENC, 1, RT, false>(ring, cst + li * BIAS_STRIDE, h, rgb0_out, rgb0_in, false, enc, sigw, st, &out);
            if (valid && h == 0) {
                // rows 0..2 of the output tile sit in accumulator registers 0..2 of lane half 0
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = P::widened_sigmoid(out[c]);
                    if (poisoned) v = __builtin_nanf("");
                    a.rgb[3 * (size_t)m + c] = v;
                }
            }
        }
        st.mark(ST_HEADS);
    }
    st.flush(a.stamps, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PREC, int WT, class RG, int ACT, bool BIGENC, int HD = 0, bool TRAIN = false>
static int launch_cfg(const MlpArgs& a, hipStream_t stream) {
    auto kern = mlp_fwd_kernel<PREC, WT, RG, ACT, BIGENC, HD, TRAIN>;
    const size_t lds_bytes = (size_t)a.const_floats * sizeof(float) + (size_t)RG::PIECES * PIECE_UNITS * UNIT_BYTES +
                             (uses_stash<PREC, WT, BIGENC>() ? (size_t)RG::WAVES * STASH_BYTES_PER_WAVE : 0);
    NERFX_REQUIRE(lds_bytes * RG::BLOCKS_PER_CU <= 160 * 1024, "model needs %zu bytes of LDS per workgroup", lds_bytes);
    NERFX_OPT_IN_LDS(kern, lds_bytes);
    int dev = 0, cus = 256;
    NERFX_HIP(hipGetDevice(&dev));
    NERFX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const uint32_t grid = std::min<uint32_t>(a.n_blocks, (uint32_t)cus * RG::BLOCKS_PER_CU);   // persistent workgroups
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * RG::WAVES), lds_bytes, stream, a);
    return launch_status("mlp_fwd_kernel");
}

template <int PREC, class RG, int ACT, bool BIGENC, bool TRAIN = false>
static int launch_width(int wt, const MlpArgs& a, hipStream_t s) {
    switch (wt) {
        case 1: return launch_cfg<PREC, 1, RG, ACT, BIGENC, 0, TRAIN>(a, s);
        case 2: return launch_cfg<PREC, 2, RG, ACT, BIGENC, 0, TRAIN>(a, s);
        case 4: return launch_cfg<PREC, 4, RG, ACT, BIGENC, 0, TRAIN>(a, s);
        case 8: return launch_cfg<PREC, 8, RG, ACT, BIGENC, 0, TRAIN>(a, s);
    }
    return fail(NERFX_E_ARG, "unsupported tile count %d", wt);
}

// Width-256 models with the skip connection and depth 4 or 8 (the reference's configurations).
template <int PREC, class RG, bool TRAIN = false, bool BIGENC = false>
static int launch_fixed_depth(int activ, int hd, const MlpArgs& a, hipStream_t s) {
#define NERFX_FIXED(ACT_)                                                                  \
    case ACT_:                                                                             \
        return hd == 2 ? launch_cfg<PREC, 8, RG, ACT_, BIGENC, 2, TRAIN>(a, s) : launch_cfg<PREC, 8, RG, ACT_, BIGENC, 4, TRAIN>(a, s);
    switch (activ) {
        NERFX_FIXED(NERFX_ACT_RELU)
        NERFX_FIXED(NERFX_ACT_LEAKY_RELU)
        NERFX_FIXED(NERFX_ACT_SILU)
    }
#undef NERFX_FIXED
    return fail(NERFX_E_ARG, "unknown activation %d", activ);
}

template <int PREC, class RG, bool BIGENC, bool TRAIN = false>
static int launch_activation(int activ, int wt, const MlpArgs& a, hipStream_t s) {
    switch (activ) {
        case NERFX_ACT_RELU: return launch_width<PREC, RG, NERFX_ACT_RELU, BIGENC, TRAIN>(wt, a, s);
        case NERFX_ACT_LEAKY_RELU: return launch_width<PREC, RG, NERFX_ACT_LEAKY_RELU, BIGENC, TRAIN>(wt, a, s);
        case NERFX_ACT_SILU: return launch_width<PREC, RG, NERFX_ACT_SILU, BIGENC, TRAIN>(wt, a, s);
    }
    return fail(NERFX_E_ARG, "unknown activation %d", activ);
}

}  // namespace nerfx
