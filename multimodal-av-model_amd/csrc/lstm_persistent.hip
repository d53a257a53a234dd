// Persistent BiLSTM layer for the bf16 perf path (H = 512): ONE launch runs all T time steps of both directions.
// 64 workgroups (2 directions x 32 tiles of 16 hidden units) stay resident; each keeps its W_hh slice (forward:
// 4 gates x 16 units x 512, backward: 16 units x 2048 of W_hh^T; 64 KiB bf16) in LDS for the whole sequence, so a
// step only moves h_{t-1} (forward) / dgates_{t+-1} (backward) through L2.  Steps are separated by a per-direction
// arrival counter: producer = every wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane
// agent-scope release + relaxed atomic add; consumer = ONE lane polls relaxed (s_sleep), agent-scope acquire,
// workgroup barrier, then plain loads (guide G16 counter form).  Spins are bounded: on timeout a flag is raised and
// the kernel still terminates.  Arithmetic, buffers and results are identical to the per-step kernels of lstm.hip
// (which remain the fp32 / generic path).
#include "av_common.h"

namespace {

constexpr int H = 512, NJT = H / 16;       // 32 unit tiles per direction
constexpr int MAXMT = 4;
constexpr int WLD = H + 8;                 // forward W slice row (bf16 elements): 64 rows x 520
constexpr int WTLD = 4 * H + 8;            // backward W^T slice row: 16 rows x 2056
constexpr unsigned SPIN_LIMIT = 1u << 18;   // ~1 us per poll: a wait that lasts a quarter of a second is a launch whose workgroups are not all resident

struct PF {
    const float* gx; const bf16_t* whh; bf16_t* hseq; float* cseq; bf16_t* gates; bf16_t* out_bt;
    int* counters;       // [2] arrivals per direction, [2] = timeout flag
    int T, B;
};
struct PB {
    const void* dout; int dout_dtype; long long do_bs, do_ts;
    bf16_t* dgates; const bf16_t* whhT; const bf16_t* gates; const float* cseq; float* dc;
    int* counters;
    int T, B;
};

// ---- exchange protocol.  AV_LSTM_SC1 = 1: the tensors other workgroups read (h_t forward, dgates_t backward) are written and
// read with agent-coherent accesses (sc0 sc1: write-through stores, cache-bypassing loads), so the hand-off needs no L2
// write-back (release fence) and no L2 invalidate (acquire fence): the producer drains its stores (vmcnt(0)) and bumps the
// counter, the consumer polls the counter and then issues the coherent loads.  AV_LSTM_SC1 = 0: plain accesses + agent fences.
#ifndef AV_LSTM_SC1
#define AV_LSTM_SC1 1
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 load16_coherent(const void* ptr) {      // the caller waits (s_waitcnt vmcnt) before using the value
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ void store4_coherent(void* ptr, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(ptr), "v"(v) : "memory");
}
// two adjacent lanes (r, r^1) hold consecutive bf16 elements: the even lane stores both as one coherent dword
__device__ __forceinline__ void store_bf16_pair_coherent(bf16_t* ptr_even_elem, float v, int r) {
    const unsigned mine = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)v);
    const unsigned other = (unsigned)__shfl_xor((int)mine, 1, 64);
    if ((r & 1) == 0) store4_coherent(ptr_even_elem, mine | (other << 16));
}

__device__ __forceinline__ void publish(int* counter, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every wave: my stores have left
    __syncthreads();
    if (tid == 0) {
#if !AV_LSTM_SC1
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void wait_for(int* counter, int target, int* flag, int tid) {
    if (tid == 0) {
        unsigned spins = 0;
        const bool dead = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // after a timeout: never spin again
        while (!dead && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
#if !AV_LSTM_SC1
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// bf16 perf path: v_exp_f32 / v_rcp_f32 forms (relative error ~1e-6, far below the bf16 the gates are stored in)
__device__ __forceinline__ float sigmoid_fast(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __expf(-2.f * fabsf(x));
    return copysignf((1.f - e) * __frcp_rn(1.f + e), x);
}
__device__ __forceinline__ void store_rest(const PF& p, int td, int row, int d, int j, float c, float h, float ig, float fg, float gg, float og) {
    const int B = p.B, T = p.T;
    p.cseq[(((long long)td * B + row) * 2 + d) * H + j] = c;
    if (p.out_bt) p.out_bt[((long long)row * T + td) * 2 * H + d * H + j] = (bf16_t)h;
    if (p.gates) {
        bf16_t* go = p.gates + (((long long)td * B + row) * 2 + d) * 4 * H;
        go[j] = (bf16_t)ig; go[H + j] = (bf16_t)fg; go[2 * H + j] = (bf16_t)gg; go[3 * H + j] = (bf16_t)og;
    }
}

#define AV_CPOL_SC0_SC1 17                   /* cache-policy operand of the LDS-DMA builtin: sc0 (bit 0) | sc1 (bit 4) */
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int ALD = H + 8;                  // forward A tile row (64 rows x 520 bf16)
constexpr int DCH = 128, NDC = 4 * H / DCH, NRING = 4, DCH_BYTES = 64 * DCH * 2;   // backward A stream: 16 chunks of 64 rows x 128 k, ring of 4

__global__ __launch_bounds__(256) void lstm_fwd_persistent(const PF p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Wl = (bf16_t*)smem;                                      // [64][WLD]: row q*16+u = gate q, unit j0+u
    bf16_t* Al = Wl + 64 * WLD;                                      // [64][ALD]: h_{t-1} rows of the current 64-row group
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int d = blockIdx.y, j0 = blockIdx.x * 16;
    const int B = p.B, T = p.T;
    const bf16_t* W = p.whh + (long long)d * 4 * H * H;
    for (int i = tid; i < 64 * (H / 8); i += 256) {                  // W_hh slice -> LDS (16 B per thread)
        const int row = i / (H / 8), ch = i % (H / 8);
        const int q = row >> 4, u = row & 15;
        *(uint4*)(Wl + row * WLD + ch * 8) = *(const uint4*)(W + (long long)(q * H + j0 + u) * H + ch * 8);
    }
    __syncthreads();
    int* cnt = p.counters + d;
    const int j = j0 + r;
    const bool defer = B <= 64;
    for (int s = 0; s < T; ++s) {
        const int td = d == 0 ? s : T - 1 - s;
        const int tp = d == 0 ? td - 1 : td + 1;
        // step-local operands do not depend on other workgroups: issue their loads BEFORE the wait so that the HBM
        // latency overlaps the hand-off (first row group; further row groups load in place)
        float pgx[4][4], pcp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = w * 16 + 4 * g + e;
            const bool ok = row < B;
            const float* gxr = p.gx + (((long long)td * B + (ok ? row : 0)) * 2 + d) * 4 * H;
#pragma unroll
            for (int q = 0; q < 4; ++q) pgx[e][q] = ok ? gxr[q * H + j] : 0.f;
            pcp[e] = (ok && s > 0) ? p.cseq[(((long long)tp * B + row) * 2 + d) * H + j] : 0.f;        // my own earlier store
        }
        if (s > 0) wait_for(cnt, NJT * s, p.counters + 2, tid);      // every tile of my direction finished step s-1
        float sv[4][6];                                              // B <= 64 (one row group): values of the deferred stores
        for (int mbase = 0; mbase < B; mbase += 64) {
            f32x4 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s > 0) {
                // h_{t-1}[mbase .. mbase+64) x 512 -> LDS, fully coalesced (one 1-KiB row per 64 lanes)
                const bf16_t* hprev = p.hseq + ((long long)tp * B) * 2 * H + d * H;
#if AV_LSTM_SC1
                if (mbase + w * 16 < B) {   // coherent LDS-DMA, one 1-KiB row per wave instruction (16 per wavefront), 16-B piece XOR-swizzled
                    // with row & 15 on the source side; rows >= B read row B-1 and are never stored.  A wavefront stages exactly the 16
                    // rows its own MFMAs read, so the hand-over is its own vmcnt wait (no workgroup barrier), and a wavefront whose
                    // row tile lies past B stages and multiplies nothing.
                    const int pos = lane;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = w * 16 + i;
                        const int rr = mbase + row < B ? mbase + row : B - 1;
                        const bf16_t* src = hprev + (long long)rr * 2 * H + ((pos ^ (row & 15)) << 3);
                        const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(row * 1024));
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)((char*)Al + off), 16, 0, AV_CPOL_SC0_SC1);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
#else
#pragma unroll
                for (int it = 0; it < 64 * (H / 8) / 256; ++it) {
                    const int i = it * 256 + tid;
                    const int row = i / (H / 8), ch = i % (H / 8);
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (mbase + row < B) v = *(const uint4*)(hprev + (long long)(mbase + row) * 2 * H + ch * 8);
                    *(uint4*)(Al + row * ALD + ch * 8) = v;
                }
                __syncthreads();
#endif
#if AV_LSTM_SC1
                const char* arow = (const char*)Al + (w * 16 + r) * 1024;      // wave w = row tile w, full K (swizzled 1-KiB rows)
#else
                const bf16_t* arow = Al + (w * 16 + r) * ALD + 8 * g;          // wave w = row tile w, full K
#endif
                if (mbase + w * 16 < B)
#pragma unroll
                for (int kk = 0; kk < H / 32; ++kk) {
#if AV_LSTM_SC1
                    const bf16x8 a = *(const bf16x8*)(arow + (((kk * 4 + g) ^ (r & 15)) << 4));
#else
                    const bf16x8 a = *(const bf16x8*)(arow + kk * 32);
#endif
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bf16x8 bq = *(const bf16x8*)(Wl + (q * 16 + r) * WLD + kk * 32 + 8 * g);
                        acc[q] = AV_MFMA_F32_16X16X32_LP(a, bq, acc[q], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = mbase + w * 16 + 4 * g + e;
                if (row < B) {
                    float gxv[4], cprev;
                    if (mbase == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) gxv[q] = pgx[e][q];
                        cprev = pcp[e];
                    } else {
                        const float* gxr = p.gx + (((long long)td * B + row) * 2 + d) * 4 * H;
#pragma unroll
                        for (int q = 0; q < 4; ++q) gxv[q] = gxr[q * H + j];
                        cprev = s > 0 ? p.cseq[(((long long)tp * B + row) * 2 + d) * H + j] : 0.f;
                    }
                    const float ig = sigmoid_fast(acc[0][e] + gxv[0]);
                    const float fg = sigmoid_fast(acc[1][e] + gxv[1]);
                    const float gg = tanh_fast(acc[2][e] + gxv[2]);
                    const float og = sigmoid_fast(acc[3][e] + gxv[3]);
                    const float c = fg * cprev + ig * gg;
                    const float h = og * tanh_fast(c);
#if AV_LSTM_SC1
                    store_bf16_pair_coherent(p.hseq + ((long long)td * B + row) * 2 * H + d * H + j, h, r);   // the only store others wait for
#else
                    p.hseq[((long long)td * B + row) * 2 * H + d * H + j] = (bf16_t)h;      // the only store other workgroups wait for
#endif
                    if (defer) { sv[e][0] = c; sv[e][1] = h; sv[e][2] = ig; sv[e][3] = fg; sv[e][4] = gg; sv[e][5] = og; }
                    else store_rest(p, td, row, d, j, c, h, ig, fg, gg, og);
                }
            }
#if AV_LSTM_SC1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my LDS reads are done before my next DMA overwrites my rows
#else
            __syncthreads();                                         // Al is rewritten by the next row group / step
#endif
        }
        if (s + 1 < T) publish(cnt, tid);
        if (defer) {                                                 // c, gates, out_bt: behind the hand-off, under the next step's wait
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = w * 16 + 4 * g + e;
                if (row < B) store_rest(p, td, row, d, j, sv[e][0], sv[e][1], sv[e][2], sv[e][3], sv[e][4], sv[e][5]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void lstm_bwd_persistent(const PB p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Wt = (bf16_t*)smem;                                      // [16][WTLD]: row u = unit j0+u, k over the 4H gate rows
    bf16_t* Ab = Wt + 16 * WTLD;                                     // ring of 4 x [64][128] chunks of dgates[t_next]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int d = blockIdx.y, j0 = blockIdx.x * 16;
    const int B = p.B, T = p.T;
    const bf16_t* W = p.whhT + (long long)d * H * 4 * H;
    for (int i = tid; i < 16 * (4 * H / 8); i += 256) {
        const int row = i / (4 * H / 8), ch = i % (4 * H / 8);
        *(uint4*)(Wt + row * WTLD + ch * 8) = *(const uint4*)(W + (long long)(j0 + row) * 4 * H + ch * 8);
    }
    __syncthreads();
    int* cnt = p.counters + d;
    const int j = j0 + r;
    for (int s = 0; s < T; ++s) {
        const int td = d == 0 ? T - 1 - s : s;
        const int tn = d == 0 ? td + 1 : td - 1;
        const int tp = d == 0 ? td - 1 : td + 1;
        float pdo[4], pg[4][4], pc[4], pcp[4], pdc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {                                // step-local operands of the first row group, before the wait
            const int row = w * 16 + 4 * g + e;
            const bool ok = row < B;
            const int rr = ok ? row : 0;
            pdo[e] = ok ? ld_any(p.dout, (long long)rr * p.do_bs + (long long)td * p.do_ts + d * H + j, p.dout_dtype) : 0.f;
            const bf16_t* gs = p.gates + (((long long)td * B + rr) * 2 + d) * 4 * H;
#pragma unroll
            for (int q = 0; q < 4; ++q) pg[e][q] = ok ? (float)gs[q * H + j] : 0.f;
            pc[e] = ok ? p.cseq[(((long long)td * B + rr) * 2 + d) * H + j] : 0.f;
            const bool has_prev = d == 0 ? td > 0 : td < T - 1;
            pcp[e] = (ok && has_prev) ? p.cseq[(((long long)tp * B + rr) * 2 + d) * H + j] : 0.f;
            pdc[e] = (ok && s > 0) ? p.dc[((long long)d * B + rr) * H + j] : 0.f;                        // my own earlier store
        }
        if (s > 0) wait_for(cnt, NJT * s, p.counters + 2, tid);
        for (int mbase = 0; mbase < B; mbase += 64) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s > 0) {
                // dh_rec = dgates[tn] x W_hh, K = 4H streamed through a 4-deep ring of 64 x 128 LDS chunks by LDS-DMA (coherent
                // loads, no registers): three chunks are always in flight (a deeper ring measured the same), so the 256 KiB a workgroup pulls per step arrive at
                // fabric bandwidth instead of one load latency per chunk.  Chunk rows are 256 B, the 16-B piece XOR-swizzled with
                // row & 15 on the source side.  Rows >= B read row B-1 (finite) and are never stored.
                const bf16_t* A = p.dgates + (((long long)tn * B) * 2 + d) * 4 * H;          // row stride 8H
                const int drow = lane >> 4, dpos = lane & 15;
                const bool has_rows = mbase + w * 16 < B;             // a row tile past B: nothing staged, nothing multiplied
                auto dma = [&](int c) {
                    char* buf = (char*)Ab + (c % NRING) * DCH_BYTES;
                    if (has_rows)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ii = w * 4 + i, row = ii * 4 + drow;
                        const int rr = mbase + row < B ? mbase + row : B - 1;
                        const bf16_t* src = A + (long long)rr * 8 * H + c * DCH + ((dpos ^ (row & 15)) << 3);
                        const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(ii * 1024));
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + off), 16, 0, AV_CPOL_SC0_SC1);
                    }
                };
                dma(0); dma(1); dma(2);
#pragma unroll
                for (int c = 0; c < NDC; ++c) {
                    // my 4 instructions of chunk c landed (the 8 of chunks c+1, c+2 may still fly).  A wavefront stages exactly the 16
                    // rows of every chunk that its own MFMAs read, so this wait is the whole hand-over: no workgroup barrier in the loop
                    if (c + 2 < NDC) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // chunks c+1, c+2 may still fly
                    else if (c + 1 < NDC) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (c + 3 < NDC) dma(c + 3);                      // my rows of ring slot (c+3) % 4 were read in iteration c-1 (lgkmcnt below)
                    const char* arow = (const char*)Ab + (c % NRING) * DCH_BYTES + (w * 16 + r) * 256;
                    const bf16_t* brow = Wt + r * WTLD + c * DCH + 8 * g;
                    if (has_rows) {
#pragma unroll
                        for (int kk = 0; kk < DCH / 32; ++kk) {
                            const bf16x8 a = *(const bf16x8*)(arow + (((kk * 4 + g) ^ (r & 15)) << 4));
                            const bf16x8 bq = *(const bf16x8*)(brow + kk * 32);
                            acc = AV_MFMA_F32_16X16X32_LP(a, bq, acc, 0, 0, 0);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads of this slot retired before iteration c+1 refills it
                    }
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = mbase + w * 16 + 4 * g + e;
                if (row < B) {
                    float dh, ig, fg, gg, og, c, cprev, dcold;
                    float* dcp = p.dc + ((long long)d * B + row) * H + j;
                    if (mbase == 0) {
                        dh = pdo[e] + acc[e]; ig = pg[e][0]; fg = pg[e][1]; gg = pg[e][2]; og = pg[e][3]; c = pc[e]; cprev = pcp[e]; dcold = pdc[e];
                    } else {
                        dh = ld_any(p.dout, (long long)row * p.do_bs + (long long)td * p.do_ts + d * H + j, p.dout_dtype) + acc[e];
                        const bf16_t* gs = p.gates + (((long long)td * B + row) * 2 + d) * 4 * H;
                        ig = (float)gs[j]; fg = (float)gs[H + j]; gg = (float)gs[2 * H + j]; og = (float)gs[3 * H + j];
                        c = p.cseq[(((long long)td * B + row) * 2 + d) * H + j];
                        const bool has_prev = d == 0 ? td > 0 : td < T - 1;
                        cprev = has_prev ? p.cseq[(((long long)tp * B + row) * 2 + d) * H + j] : 0.f;
                        dcold = s > 0 ? *dcp : 0.f;
                    }
                    const float tc = tanh_fast(c);
                    const float dcs = dcold + dh * og * (1.f - tc * tc);
                    *dcp = dcs * fg;
                    bf16_t* dg = p.dgates + (((long long)td * B + row) * 2 + d) * 4 * H;
#if AV_LSTM_SC1
                    store_bf16_pair_coherent(dg + j, dcs * gg * ig * (1.f - ig), r);
                    store_bf16_pair_coherent(dg + H + j, dcs * cprev * fg * (1.f - fg), r);
                    store_bf16_pair_coherent(dg + 2 * H + j, dcs * ig * (1.f - gg * gg), r);
                    store_bf16_pair_coherent(dg + 3 * H + j, dh * tc * og * (1.f - og), r);
#else
                    dg[j] = (bf16_t)(dcs * gg * ig * (1.f - ig));
                    dg[H + j] = (bf16_t)(dcs * cprev * fg * (1.f - fg));
                    dg[2 * H + j] = (bf16_t)(dcs * ig * (1.f - gg * gg));
                    dg[3 * H + j] = (bf16_t)(dh * tc * og * (1.f - og));
#endif
                }
            }
            __syncthreads();
        }
        if (s + 1 < T) publish(cnt, tid);
    }
}

// ---- row-split variant (default): grid = 32 unit tiles x 2 directions x NRG row groups (NRG = min(ceil(B / 16), 4)).  Batch rows never
// interact in the recurrence, so every (direction, row group) is its own chain of 32 workgroups with its own arrival counter (one
// 128-B line each).  A workgroup owns 16 hidden units x the 16-row tiles rt = z, z + NRG, ...; its W_hh slice lives in REGISTERS
// (64 VGPRs per lane: wavefront w holds the MFMA B fragments of its quarter of K), so a step stages only the 16 x K operand rows in
// one shot of wave-private LDS-DMA (forward 4, backward 16 instructions per wavefront, all in flight at once, no ring, no workgroup
// barrier before the MFMAs), multiplies its K quarter (forward 16, backward 16 MFMAs per wavefront instead of 64) and the four
// partial tiles meet in LDS; the element-wise tail runs one (row, unit) per thread.  Per-CU traffic per step drops 4x (backward:
// 64 KiB instead of 256 KiB) and 256 CUs pull instead of 64.  Same arithmetic order inside a K quarter; the cross-quarter sum is
// ((q0 + q1) + q2) + q3 in fp32 (the 64-row kernel accumulates the same products in one chain: results agree to fp32 rounding).
constexpr int CNT_STRIDE = 32;               // ints between counters
constexpr int MAXRG = 8;                     // row groups per direction at most (two workgroups per CU on 256 CUs)
constexpr int PSZ = 320;                     // floats of one padded 16 x 16 partial tile
__device__ __forceinline__ int* group_counter(int* counters, int d, int z) { return counters + CNT_STRIDE * (1 + d * MAXRG + z); }
__device__ __forceinline__ int pidx(int row, int unit) { return (row >> 2) * 80 + (row & 3) * 16 + unit; }   // conflict-free for the C layout

__global__ __launch_bounds__(256, 2) void lstm_fwd_split(const PF p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Al = smem;                                                 // [4 wavefronts][16 rows][256 B]: h_{t-1}, K quarter per wavefront
    float* part = (float*)(smem + 16384);                            // [4 wavefronts][4 gates][PSZ]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int d = blockIdx.y, j0 = blockIdx.x * 16, z = blockIdx.z, nrg = gridDim.z;
    const int B = p.B, T = p.T, ntile = (B + 15) / 16;
    const bf16_t* W = p.whh + (long long)d * 4 * H * H;
    bf16x8 wf[4][4];                                                 // [gate][kk]: B fragments, k = w*128 + kk*32 + 8g .. +7, column = unit j0 + r
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wf[q][kk] = *(const bf16x8*)(W + (long long)(q * H + j0 + r) * H + w * 128 + kk * 32 + 8 * g);
    int* cnt = group_counter(p.counters, d, z);
    const int erow = tid >> 4, eu = tid & 15, j = j0 + eu;           // element-wise tail: one (row, unit) per thread
    const bool defer = z + nrg >= ntile;                             // one row tile per workgroup: stores nobody waits for go behind the hand-off
    for (int s = 0; s < T; ++s) {
        const int td = d == 0 ? s : T - 1 - s;
        const int tp = d == 0 ? td - 1 : td + 1;
        float pgx[4], pcp;
        {   // operands that do not depend on other workgroups: loaded BEFORE the wait (first row tile)
            const int row = z * 16 + erow;
            const bool ok = row < B;
            const float* gxr = p.gx + (((long long)td * B + (ok ? row : 0)) * 2 + d) * 4 * H;
#pragma unroll
            for (int q = 0; q < 4; ++q) pgx[q] = ok ? gxr[q * H + j] : 0.f;
            pcp = (ok && s > 0) ? p.cseq[(((long long)tp * B + row) * 2 + d) * H + j] : 0.f;             // my own earlier store
        }
        if (s > 0) wait_for(cnt, NJT * s, p.counters + 2, tid);      // every unit tile of my (direction, row group) finished step s-1
        float sv[6];
        for (int rt = z; rt < ntile; rt += nrg) {
            if (s > 0) {
                const bf16_t* hprev = p.hseq + ((long long)tp * B) * 2 * H + d * H;
#pragma unroll
                for (int i = 0; i < 4; ++i) {                        // one instruction = 4 rows x 256 B; the 16-B piece XOR-swizzled with row & 15 on the source side
                    const int row = 4 * i + g;
                    const int rr = rt * 16 + row < B ? rt * 16 + row : B - 1;                           // rows >= B: finite duplicates, never stored
                    const bf16_t* src = hprev + (long long)rr * 2 * H + w * 128 + ((r ^ row) << 3);
                    const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(w * 4096 + i * 1024));
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Al + off), 16, 0, AV_CPOL_SC0_SC1);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                f32x4 acc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                const char* arow = Al + w * 4096 + r * 256;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bf16x8 a = *(const bf16x8*)(arow + (((kk * 4 + g) ^ r) << 4));
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = AV_MFMA_F32_16X16X32_LP(a, wf[q][kk], acc[q], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) part[(w * 4 + q) * PSZ + pidx(4 * g + e, r)] = acc[q][e];
                __syncthreads();
            }
            const int row = rt * 16 + erow;
            if (row < B) {
                float gxv[4], cprev;
                if (rt == z) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) gxv[q] = pgx[q];
                    cprev = pcp;
                } else {
                    const float* gxr = p.gx + (((long long)td * B + row) * 2 + d) * 4 * H;
#pragma unroll
                    for (int q = 0; q < 4; ++q) gxv[q] = gxr[q * H + j];
                    cprev = s > 0 ? p.cseq[(((long long)tp * B + row) * 2 + d) * H + j] : 0.f;
                }
                if (s > 0) {
                    const int pi = pidx(erow, eu);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        gxv[q] += ((part[q * PSZ + pi] + part[(4 + q) * PSZ + pi]) + part[(8 + q) * PSZ + pi]) + part[(12 + q) * PSZ + pi];
                }
                const float ig = sigmoid_fast(gxv[0]);
                const float fg = sigmoid_fast(gxv[1]);
                const float gg = tanh_fast(gxv[2]);
                const float og = sigmoid_fast(gxv[3]);
                const float c = fg * cprev + ig * gg;
                const float h = og * tanh_fast(c);
                store_bf16_pair_coherent(p.hseq + ((long long)td * B + row) * 2 * H + d * H + j, h, eu);       // the only store others wait for
                if (defer) { sv[0] = c; sv[1] = h; sv[2] = ig; sv[3] = fg; sv[4] = gg; sv[5] = og; }
                else store_rest(p, td, row, d, j, c, h, ig, fg, gg, og);
            }
            if (rt + nrg < ntile) __syncthreads();                   // the partial tiles are rewritten by the next row tile
        }
        if (s + 1 < T) publish(cnt, tid);
        if (defer) {
            const int row = z * 16 + erow;
            if (row < B) store_rest(p, td, row, d, j, sv[0], sv[1], sv[2], sv[3], sv[4], sv[5]);
        }
    }
}

__global__ __launch_bounds__(256, 2) void lstm_bwd_split(const PB p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ab = smem;                                                 // [4 wavefronts][16 rows][1024 B]: dgates[t_next], K quarter per wavefront
    float* part = (float*)(smem + 65536);                            // [4 wavefronts][PSZ]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int d = blockIdx.y, j0 = blockIdx.x * 16, z = blockIdx.z, nrg = gridDim.z;
    const int B = p.B, T = p.T, ntile = (B + 15) / 16;
    const bf16_t* W = p.whhT + (long long)d * H * 4 * H;
    bf16x8 wt[16];                                                   // B fragments of W_hh^T: k = w*512 + kk*32 + 8g .. +7 (gate rows), column = unit j0 + r
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) wt[kk] = *(const bf16x8*)(W + (long long)(j0 + r) * 4 * H + w * 512 + kk * 32 + 8 * g);
    int* cnt = group_counter(p.counters, d, z);
    const int erow = tid >> 4, eu = tid & 15, j = j0 + eu;
    for (int s = 0; s < T; ++s) {
        const int td = d == 0 ? T - 1 - s : s;
        const int tn = d == 0 ? td + 1 : td - 1;
        const int tp = d == 0 ? td - 1 : td + 1;
        const bool has_prev = d == 0 ? td > 0 : td < T - 1;
        float pdo, pg[4], pc, pcp, pdc;
        {   // step-local operands of the first row tile, before the wait
            const int row = z * 16 + erow;
            const bool ok = row < B;
            const int rr = ok ? row : 0;
            pdo = ok ? ld_any(p.dout, (long long)rr * p.do_bs + (long long)td * p.do_ts + d * H + j, p.dout_dtype) : 0.f;
            const bf16_t* gs = p.gates + (((long long)td * B + rr) * 2 + d) * 4 * H;
#pragma unroll
            for (int q = 0; q < 4; ++q) pg[q] = ok ? (float)gs[q * H + j] : 0.f;
            pc = ok ? p.cseq[(((long long)td * B + rr) * 2 + d) * H + j] : 0.f;
            pcp = (ok && has_prev) ? p.cseq[(((long long)tp * B + rr) * 2 + d) * H + j] : 0.f;
            pdc = (ok && s > 0) ? p.dc[((long long)d * B + rr) * H + j] : 0.f;                           // my own earlier store
        }
        if (s > 0) wait_for(cnt, NJT * s, p.counters + 2, tid);
        for (int rt = z; rt < ntile; rt += nrg) {
            if (s > 0) {
                const bf16_t* A = p.dgates + (((long long)tn * B) * 2 + d) * 4 * H;                      // row stride 8H
#pragma unroll
                for (int i = 0; i < 16; ++i) {                       // one instruction = one row's K quarter (1 KiB), piece XOR-swizzled with row & 15
                    const int rr = rt * 16 + i < B ? rt * 16 + i : B - 1;
                    const bf16_t* src = A + (long long)rr * 8 * H + w * 512 + ((lane ^ i) << 3);
                    const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(w * 16384 + i * 1024));
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ab + off), 16, 0, AV_CPOL_SC0_SC1);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                const char* arow = Ab + w * 16384 + r * 1024;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const bf16x8 a = *(const bf16x8*)(arow + (((kk * 4 + g) ^ r) << 4));
                    acc = AV_MFMA_F32_16X16X32_LP(a, wt[kk], acc, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) part[w * PSZ + pidx(4 * g + e, r)] = acc[e];
                __syncthreads();
            }
            const int row = rt * 16 + erow;
            if (row < B) {
                float dh, ig, fg, gg, og, c, cprev, dcold;
                float* dcp = p.dc + ((long long)d * B + row) * H + j;
                if (rt == z) {
                    dh = pdo; ig = pg[0]; fg = pg[1]; gg = pg[2]; og = pg[3]; c = pc; cprev = pcp; dcold = pdc;
                } else {
                    dh = ld_any(p.dout, (long long)row * p.do_bs + (long long)td * p.do_ts + d * H + j, p.dout_dtype);
                    const bf16_t* gs = p.gates + (((long long)td * B + row) * 2 + d) * 4 * H;
                    ig = (float)gs[j]; fg = (float)gs[H + j]; gg = (float)gs[2 * H + j]; og = (float)gs[3 * H + j];
                    c = p.cseq[(((long long)td * B + row) * 2 + d) * H + j];
                    cprev = has_prev ? p.cseq[(((long long)tp * B + row) * 2 + d) * H + j] : 0.f;
                    dcold = s > 0 ? *dcp : 0.f;
                }
                if (s > 0) {
                    const int pi = pidx(erow, eu);
                    dh += ((part[pi] + part[PSZ + pi]) + part[2 * PSZ + pi]) + part[3 * PSZ + pi];
                }
                const float tc = tanh_fast(c);
                const float dcs = dcold + dh * og * (1.f - tc * tc);
                *dcp = dcs * fg;
                bf16_t* dg = p.dgates + (((long long)td * B + row) * 2 + d) * 4 * H;
                store_bf16_pair_coherent(dg + j, dcs * gg * ig * (1.f - ig), eu);
                store_bf16_pair_coherent(dg + H + j, dcs * cprev * fg * (1.f - fg), eu);
                store_bf16_pair_coherent(dg + 2 * H + j, dcs * ig * (1.f - gg * gg), eu);
                store_bf16_pair_coherent(dg + 3 * H + j, dh * tc * og * (1.f - og), eu);
            }
            if (rt + nrg < ntile) __syncthreads();
        }
        if (s + 1 < T) publish(cnt, tid);
    }
}

constexpr int LDS_FS = 16384 + 16 * PSZ * 4;                       // 36 864
constexpr int LDS_BS = 65536 + 4 * PSZ * 4;                        // 70 656

constexpr int LDS_F = 64 * WLD * 2 + 64 * ALD * 2;            // 66 560 + 66 560
constexpr int LDS_B = 16 * WTLD * 2 + NRING * DCH_BYTES;         // 65 792 + 65 536

}  // namespace

static bool lstm_split_enabled() {
    static const bool on = [] { const char* e = getenv("AVAMD_LSTM_SPLIT"); return !(e && e[0] == '0'); }();
    return on;
}
// row groups that can be co-resident: every workgroup of the launch spins on its peers, so the whole grid (64 x row groups) must fit
// on the device at two workgroups per CU (a partitioned GPU exposes fewer CUs)
static int lstm_prepare(void);
static int lstm_row_groups(int B) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    }
    const int ntile = (B + 15) / 16;
    // up to 8 row groups = 512 workgroups, two per CU (forward 36 KB, backward 71 KB of LDS each): at B = 128 (both speakers in one call) every
    // workgroup then has ONE 16-row tile per step instead of two in sequence.  AVAMD_LSTM_NRG caps it (4 = the round-2 geometry).
    static const int cap = [] { const char* e = getenv("AVAMD_LSTM_NRG"); const int v = e ? atoi(e) : MAXRG; return v < 1 ? 1 : (v > MAXRG ? MAXRG : v); }();
    // more than 4 row groups need TWO co-resident workgroups per CU: ask the runtime once what the split kernels' registers and LDS really
    // admit on this device / partition (the count alone was an inference); fewer than two -> the round-2 geometry (one per CU)
    static int per_cu = 0;
    if (!per_cu) {
        int f = 0, b = 0;
        if (lstm_prepare() != AV_OK ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&f, (const void*)lstm_fwd_split, 256, LDS_FS) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (const void*)lstm_bwd_split, 256, LDS_BS) != hipSuccess) { f = b = 1; (void)hipGetLastError(); }
        per_cu = f < b ? f : b;
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 2) per_cu = 2;
    }
    int nrg = ntile < cap ? ntile : cap;
    while (nrg > 1 && 2 * NJT * nrg > per_cu * ncu) --nrg;
    return 2 * NJT * nrg <= per_cu * ncu ? nrg : 0;
}

static int lstm_prepare(void) {
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute((const void*)lstm_fwd_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_F) != hipSuccess ||
            hipFuncSetAttribute((const void*)lstm_bwd_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B) != hipSuccess ||
            hipFuncSetAttribute((const void*)lstm_fwd_split, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FS) != hipSuccess ||
            hipFuncSetAttribute((const void*)lstm_bwd_split, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BS) != hipSuccess) {
            av_set_error("av_lstm_*_layer: cannot raise dynamic LDS"); return AV_ERR_LAUNCH;
        }
        done = true;
    }
    return AV_OK;
}

// counters: AV_LSTM_COUNTER_INTS (1024) ints of workspace (zeroed here); returns after enqueueing; the timeout flag counters[2] can be read later
extern "C" int av_lstm_fwd_layer(const float* gx, const void* whh, void* hseq, float* cseq, void* gates, void* out_bt, int* counters,
                                 int T, int B, int Hh, void* stream) {
    AV_CHECK(gx && whh && hseq && cseq && counters, "av_lstm_fwd_layer: null pointer");
    AV_CHECK(Hh == H && T > 0 && B > 0, "av_lstm_fwd_layer: persistent kernel is built for H=512 (got %d), T=%d B=%d", Hh, T, B);
    hipStream_t st = (hipStream_t)stream;
    if (lstm_prepare() != AV_OK) return AV_ERR_LAUNCH;
    if (hipMemsetAsync(counters, 0, AV_LSTM_COUNTER_INTS * sizeof(int), st) != hipSuccess) { av_set_error("av_lstm_fwd_layer: memset failed"); return AV_ERR_LAUNCH; }
    PF p{gx, (const bf16_t*)whh, (bf16_t*)hseq, cseq, (bf16_t*)gates, (bf16_t*)out_bt, counters, T, B};
    const int nrg = lstm_row_groups(B);
    AV_CHECK(nrg > 0, "av_lstm_fwd_layer: the persistent kernel needs 32 compute units (use the step kernels)");
    if (lstm_split_enabled()) {
        hipLaunchKernelGGL(lstm_fwd_split, dim3(NJT, 2, nrg), dim3(256), LDS_FS, st, p);
    } else {
        hipLaunchKernelGGL(lstm_fwd_persistent, dim3(NJT, 2), dim3(256), LDS_F, st, p);
    }
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_lstm_bwd_layer(const void* dout, int dout_dtype, long long do_bs, long long do_ts, void* dgates, const void* whhT,
                                 const void* gates, const float* cseq, float* dc, int* counters, int T, int B, int Hh, void* stream) {
    AV_CHECK(dout && dgates && whhT && gates && cseq && dc && counters, "av_lstm_bwd_layer: null pointer");
    AV_CHECK(Hh == H && T > 0 && B > 0, "av_lstm_bwd_layer: persistent kernel is built for H=512 (got %d), T=%d B=%d", Hh, T, B);
    hipStream_t st = (hipStream_t)stream;
    if (lstm_prepare() != AV_OK) return AV_ERR_LAUNCH;
    if (hipMemsetAsync(counters, 0, AV_LSTM_COUNTER_INTS * sizeof(int), st) != hipSuccess) { av_set_error("av_lstm_bwd_layer: memset failed"); return AV_ERR_LAUNCH; }
    PB p{dout, dout_dtype, do_bs, do_ts, (bf16_t*)dgates, (const bf16_t*)whhT, (const bf16_t*)gates, cseq, dc, counters, T, B};
    const int nrg = lstm_row_groups(B);
    AV_CHECK(nrg > 0, "av_lstm_bwd_layer: the persistent kernel needs 32 compute units (use the step kernels)");
    if (lstm_split_enabled()) {
        hipLaunchKernelGGL(lstm_bwd_split, dim3(NJT, 2, nrg), dim3(256), LDS_BS, st, p);
    } else {
        hipLaunchKernelGGL(lstm_bwd_persistent, dim3(NJT, 2), dim3(256), LDS_B, st, p);
    }
    AV_LAUNCH_CHECK();
    return AV_OK;
}
