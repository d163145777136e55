// Lab kernel (not product code), round 5: the persistent encoder GEMM of csrc/gemm_f16p.hip with K64 SLOTS -- the LDS image has rows of
// 128 bytes, so every LDS-DMA piece is 8 rows x one FULL 128-byte cache line instead of 16 rows x half a line (the product kernel's K32
// stages fetch every line of A and W in two requests a stage apart: profiles/r5ai_*: TCP_TCC_READ_REQ = one request per 64 bytes).  Two slots
// of 64 KB; slot g + 1 is requested while the first half of slot g is multiplied.  Same MFMA sequence per output element, same epilogues.
#include <type_traits>
#ifndef F16R_ISSUE_ALL_L1
#define F16R_ISSUE_ALL_L1 0      // 1: all eight pieces of the next slot in the first load interval of a slot (default: four there, four in the second)
#endif
#ifndef F16R_EPI_ISSUE
#define F16R_EPI_ISSUE 1         // 1: the next tile's SECOND slot is requested at the head of the epilogue (ahead of its stores), so that the first wait of the
#endif                           //    next tile can leave the stores in flight; 0: every slot is requested while its predecessor's first half is multiplied
#ifndef F16R_EPI_SPLIT
#define F16R_EPI_SPLIT 0         // 1: residual epilogue in two phases -- everything in front of the residual add for the whole tile, then add + store
#endif
#ifndef F16R_FULL_LINE_EPI
#define F16R_FULL_LINE_EPI 0     // 1: the SIMPLE epilogue's stores (and residual loads) cover WHOLE 128-byte lines: 8 rows x 128 B per instruction instead of
#endif                           //    16 rows x 64 B -- the two 16-byte chunks a lane holds of a row are exchanged with lane ^ 8 (DPP row_ror:8) first
#ifndef F16R_NT_EPI
#define F16R_NT_EPI 0            // bit 0: the epilogue's C stores non-temporal; bit 1: its residual loads non-temporal (both are touched once by this kernel)
#endif
#ifndef F16R_RES_EARLY
#define F16R_RES_EARLY 0         // n = 1..4: the first n (of 4) row blocks of the tile's residual rows are REQUESTED (into the epilogue's own registers) while the
#endif                           //    tile's last slot is multiplied, instead of at the head of the epilogue
#ifndef F16R_RES_PREFETCH
#define F16R_RES_PREFETCH 0      // 1: the tile's residual lines are touched (one dword per line, result unused) while its last slot is multiplied
#endif
namespace wm {

namespace f16r {
constexpr int BM = 256, BN = 256, BK = 32, NWAVE = 8;
constexpr int A_PART = BM * 64 * 2, SLOT = (BM + BN) * 64 * 2;      // a K64 slot: 32 KB + 32 KB, rows of 128 B (full cache lines)

constexpr int MAX_N = 8192;                                          // the bias vector sits in LDS behind the ring: 16 KB
}  // namespace f16r

// SIMPLE: plain row-major output and residual (the encoder layers' four GEMMs): the epilogue then carries none of the
// strided-view / head-split / int8 address arithmetic (integer divisions, their branches) the general form is compiled with.
template <int STAGES, int ACT, bool SIMPLE = false>      // ACT: 0 none, 1 erf-GELU, 2 tanh-GELU
__global__ __launch_bounds__(512) void gemm_f16r_kernel(GemmBigParams p) {
    using namespace f16r;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    // store instructions of one wave's epilogue, the number the first stage wait of the next tile may leave in flight ON TOP of the DMA
    // pieces: 16 stores of 16 bytes in the SIMPLE form (4 row blocks x 4 block pairs), 32 of 8 bytes in the general one.  It must not
    // exceed what the epilogue really issues: a larger count would let the wait pass with DMA pieces of the stage still in flight.
    constexpr int N_STORES = SIMPLE ? 16 : 32;

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;           // 4 (M) x 2 (N) waves, each 64 rows x 128 channels
    const int g = lane >> 4;

    // ---- this workgroup's tiles: XCD x owns the contiguous band [lo, hi) of the tile list (channel tile fastest) -----
    const int nt_n = p.N / BN, nt_m = (p.M + BM - 1) / BM, n_tiles = nt_n * nt_m;
    const int per_xcd = gridDim.x >> 3;              // gridDim.x is a multiple of 8
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3;
    const int band = (n_tiles + 7) >> 3;
    const int lo = min(n_tiles, xcd * band), hi = min(n_tiles, lo + band);
    const int my_tiles = (hi - lo - j0 + per_xcd - 1) > 0 ? (hi - lo - j0 + per_xcd - 1) / per_xcd : 0;
    if (my_tiles == 0) return;
    const int nk = p.K / 64;                         // K64 slots per tile
    // Order of a band's tiles.  In plain row-major order the workgroups of an XCD, which march over K roughly in step, hold
    // ~32 / nt_n row panels against ALL nt_n channel tiles: each A panel is fetched once, but the whole weight matrix streams
    // through the XCD's 4 MB L2 once per row panel (N = 5120, K = 1280: 13 MB of W per 0.66 MB of A -- the kernel fetched 11x
    // its algorithmic bytes, 8.4 GB per launch at M = 288 000, 2 TB/s of fabric traffic that the decode loop beside it pays
    // for: profiles/r3g_pmc_stage_b192_pass2.txt).  Instead the band's full tile rows are taken R at a time (R row panels =
    // <= 2.75 MB stay in L2), channel tile by channel tile: the weights stream past R panels at once -- 1 / R of the traffic.
    // Head and tail of the band (partial tile rows) keep the plain order.  Same tiles, same arithmetic; only the order changes.
    const int R = p.tile_rows > 0 ? p.tile_rows : max(1, min(8, (int)((2816u << 10) / ((unsigned)p.K * BM * 2u))));
    const int row_first = (lo + nt_n - 1) / nt_n, row_last = hi / nt_n;              // full tile rows [row_first, row_last)
    const int q_head = min(hi, row_first * nt_n) - lo, q_mid = max(0, row_last - row_first) * nt_n;
    auto tile_of = [&](int q, int& tm, int& tn) {     // q-th tile of the band -> (row panel, channel tile)
        if (q < q_head) { const int tile = lo + q; tm = tile / nt_n; tn = tile - tm * nt_n; return; }
        const int q2 = q - q_head;
        if (q2 >= q_mid) { const int tile = row_last * nt_n + (q2 - q_mid); tm = tile / nt_n; tn = tile - tm * nt_n; return; }
        const int sr = q2 / (R * nt_n), rem = q2 - sr * (R * nt_n);
        const int rows_here = min(R, (row_last - row_first) - sr * R);
        tn = rem / rows_here;
        tm = row_first + sr * R + (rem - tn * rows_here);
    };

    // ---- loader: per K64 slot this wave requests 4 A pieces and 4 W pieces of 8 rows x 128 B (FULL cache lines) ----------------
    const unsigned char* a_base = nullptr;
    const unsigned char* w_base = nullptr;
    uint32_t a_lane[4], w_lane0;
    auto row_offset = [&](int gr) -> size_t {
        return p.a_rows > 0 ? (size_t)(gr / p.a_rows) * p.a_bstride + (size_t)(gr % p.a_rows) * p.lda : (size_t)gr * p.lda;
    };
    {
        const int r = wid * 8 + (lane >> 3);                                      // row inside the 256-row part (piece q: + 64 q rows, same swizzle)
        const int c = (lane & 7) ^ ((r >> 1) & 7);                                // source chunk for this LDS slot
        w_lane0 = (uint32_t)r * (uint32_t)p.K * 2u + c * 16;
    }
    auto set_tile = [&](int t) {
        int tm, tn;
        tile_of(j0 + t * per_xcd, tm, tn);
        const size_t off0 = row_offset(tm * BM);
        a_base = (const unsigned char*)(p.A + off0);
        w_base = (const unsigned char*)p.W + (size_t)tn * BN * p.K * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (wid + NWAVE * q) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gr = tm * BM + r;
            if (gr > p.M - 1) gr = p.M - 1;
            a_lane[q] = (uint32_t)((row_offset(gr) - off0) * 2) + c * 16;
        }
    };
    int load_ks = 0, load_tile = 0;
    uint32_t load_slot = 0;
    auto issue_pair = [&](int q0) {                  // pieces q0 and q0 + 1 of A and of W of the slot being requested
#pragma unroll
        for (int q = q0; q < q0 + 2; ++q) {
            unsigned char* dst = smem + load_slot + (wid + NWAVE * q) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + a_lane[q]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_base + (size_t)q * 64 * p.K * 2 + w_lane0),
                                             (__attribute__((address_space(3))) void*)(dst + A_PART), 16, 0, 0);
        }
        if (q0 == 2) {
            a_base += 128; w_base += 128;
            load_slot ^= SLOT;
            if (++load_ks == nk) { load_ks = 0; ++load_tile; set_tile(load_tile < my_tiles ? load_tile : my_tiles - 1); }
        }
    };
    set_tile(0);
    issue_pair(0); issue_pair(2);                    // slot 0 <- the first K64 of the first tile
    if (F16R_EPI_ISSUE) { issue_pair(0); issue_pair(2); }      // ... and slot 1 <- its second

    // ---- fragment addresses inside a slot (rows of 128 B; chunk c of row r at position c ^ ((r >> 1) & 7)) ---------------------
    const int sw = ((lane & 15) >> 1) & 7;
    const int a_off0 = (wr * 64 + (lane & 15)) * 128, b_off0 = A_PART + (wc * 128 + (lane & 15)) * 128;
    const int ch0 = (g ^ sw) << 4;                   // the second 64-byte half: ^ 64

    float4v acc[4][8];
    half8v af[4], bx[4];                             // A rows (4 blocks); W channels, first or second 64 of this wave (4 blocks)

    // the whole bias vector (N <= 8192 channels, zeros without one) behind the ring: the epilogues read it with LDS loads,
    // which do not go through the global-memory counter the DMA pipeline and the stores are timed with
    {
        h16* bias_lds0 = (h16*)(smem + 2 * SLOT);
        for (int c = tid; c < p.N; c += 512) bias_lds0[c] = p.bias ? p.bias[c] : (h16)0.f;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // The two waves of a SIMD (w and w + 4) alternate: while one multiplies (16 MFMAs, nothing else), the other reads its next
    // fragments from LDS, requests DMA pieces and waits -- the matrix pipe always has one wave feeding it.  The workgroup's
    // barriers are the clock of that alternation; waves 4-7 run one barrier behind waves 0-3.
    if (wid >= 4) __builtin_amdgcn_s_barrier();

    uint32_t cons_slot = 0;                          // ring slot (byte offset) of the stage being multiplied
    int t = 0;
    bool after_epilogue = false;                     // the first wait of the tile has this wave's epilogue stores behind the pieces it waits for
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    bool prefetch_now = false;
    uint4 r8e[F16R_RES_EARLY ? F16R_RES_EARLY : 1][4];
    auto res_early = [&]() {                         // (same addresses as finish_simple's loads)
        int tm, tn;
        tile_of(j0 + t * per_xcd, tm, tn);
        const int ge_ = lane >> 4, rl_ = lane & 15, odd_ = ge_ & 1;
        const int colx_ = tn * BN + wc * 128 + 4 * (ge_ - odd_) + 16 * odd_;
#pragma unroll
        for (int i = 0; i < (F16R_RES_EARLY ? F16R_RES_EARLY : 1); ++i) {
            const int row = tm * BM + wr * 64 + i * 16 + rl_;
            const h16* rrow = p.residual + (size_t)(row < p.M ? row : p.M - 1) * p.ldr + colx_;
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) r8e[i][jp] = *(const uint4*)(rrow + jp * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    unsigned pf_dummy0 = 0, pf_dummy1 = 0;
    auto res_prefetch = [&]() {                      // this wave's 64 rows x 256 B of the residual tile: lane = row, two lines per row
        int tm, tn;
        tile_of(j0 + t * per_xcd, tm, tn);
        int row = tm * BM + wr * 64 + lane;
        if (row > p.M - 1) row = p.M - 1;
        const h16* src = p.residual + (size_t)row * p.ldr + tn * BN + wc * 128;
        asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %2, off offset:128" : "=&v"(pf_dummy0), "=&v"(pf_dummy1) : "v"(src) : "memory");
    };
    auto stage = [&](auto half_tag, auto first_tag) {
        constexpr int HALF = decltype(half_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;      // the first slot of a tile: with F16R_EPI_ISSUE its successor was requested by the epilogue before          // which 64-byte half of the slot's rows
        const unsigned char* st = smem + cons_slot;
        const int ch = HALF ? (ch0 ^ 64) : ch0;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const half8v*)(st + a_off0 + ch + i * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) bx[j] = *(const half8v*)(st + b_off0 + ch + j * 2048);
        if (HALF == 0 && !(FIRST && F16R_EPI_ISSUE)) { issue_pair(0); if (F16R_ISSUE_ALL_L1) issue_pair(2); }
        if (HALF == 0 && F16R_RES_PREFETCH && SIMPLE) { if (prefetch_now) res_prefetch(); }
        if (HALF == 0 && F16R_RES_EARLY && SIMPLE && ACT == 0) { if (prefetch_now) res_early(); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bx[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) bx[j] = *(const half8v*)(st + b_off0 + ch + (4 + j) * 2048);
        if (HALF == 0) { if (!F16R_ISSUE_ALL_L1 && !(FIRST && F16R_EPI_ISSUE)) issue_pair(2); }
        else if (FIRST && F16R_EPI_ISSUE && after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_STORES) : "memory");      // (the slot's pieces are OLDER than the stores)
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (F16R_RES_PREFETCH) asm volatile("" :: "v"(pf_dummy0), "v"(pf_dummy1)); }   // the other slot (requested a sub-stage ago) has landed before the next barrier
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bx[j], af[i], acc[i][4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (HALF == 1) cons_slot ^= SLOT;
    };
    for (;;) {
        stage(std::integral_constant<int, 0>{}, std::true_type{}); stage(std::integral_constant<int, 1>{}, std::true_type{});
        after_epilogue = false;
#pragma unroll 1
        for (int ks = 1; ks < nk; ++ks) {
            if ((F16R_RES_PREFETCH || F16R_RES_EARLY) && SIMPLE) prefetch_now = (ks == nk - 1) && p.residual != nullptr;
            stage(std::integral_constant<int, 0>{}, std::false_type{}); stage(std::integral_constant<int, 1>{}, std::false_type{});
        }

        // Waves 4-7 run one barrier behind: their last barrier of the tile pairs with THIS one.  Without it (round 2) it paired with
        // waves 0-3's first barrier of the next tile, i.e. waves 4-7 sat behind their finished last multiply until waves 0-3 had
        // run their whole epilogue, and waves 0-3 then waited at their next barrier for waves 4-7's epilogue: the two epilogues ran
        // one after the other -- 10 us of a plain K = 1280 tile's 47, 17-21 us with a residual (time stamps inside the kernel,
        // profiles/r3w_gemm_tile_boundary_stamps_before.log).  Now both groups enter their epilogues together; waves 4-7 take their
        // lag back with an extra barrier at the start of the next tile (below), as at the start of the kernel.
        if (wid < 4) __builtin_amdgcn_s_barrier();

        if (F16R_EPI_ISSUE) { issue_pair(0); issue_pair(2); __builtin_amdgcn_sched_barrier(0); }      // the slot just multiplied <- the next tile's second K64
        // ================================ epilogue (as gemm_f16.hip) =====================================================
        int tm, tn;
        tile_of(j0 + t * per_xcd, tm, tn);
        const int row0 = tm * BM, col0 = tn * BN;
        const int hs_b0 = !SIMPLE && p.out_mode == 1 ? row0 / p.hs_T : 0, hs_t0 = !SIMPLE && p.out_mode == 1 ? row0 - hs_b0 * p.hs_T : 0;     // wave-uniform
        // every lane-dependent quantity of the epilogue is derived from `le`, which the compiler cannot see through: nothing
        // of the epilogue's address arithmetic is hoisted out of the tile loop into registers the K loop needs
        int le = lane;
        asm volatile("" : "+v"(le));
        const int ge = le >> 4, rl = le & 15;
        const bool scale_cols = p.colscale_n > 0;
        const int colw = col0 + wc * 128 + ge * 4;                      // this lane's first column
        // Every global LOAD of the epilogue precedes every STORE: the memory counter is in order, so a load requested behind a
        // piece's stores could only be waited for together with them.  The bias comes from LDS (copied there once per launch),
        // the residual rows (64 registers) are all requested up front; then nothing but arithmetic and stores, piece by piece
        // (one piece = one 16-row block x 64 channels: few temporaries alive at a time).
        const h16* bias_lds = (const h16*)(smem + 2 * SLOT) + col0 + wc * 128 + ge * 4;
        auto finish = [&](auto res_tag) {
            constexpr bool RES = decltype(res_tag)::value;
            half4v r4[RES ? 4 : 1][RES ? 8 : 1];
            if constexpr (RES) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = row0 + wr * 64 + i * 16 + rl;
                    const int rowc = row < p.M ? row : p.M - 1;
                    const h16* rrow = p.residual + (size_t)(!SIMPLE && p.res_mod > 0 ? rowc % p.res_mod : rowc) * p.ldr + colw;
#pragma unroll
                    for (int j = 0; j < 8; ++j) r4[i][j] = *(const half4v*)(rrow + j * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + wr * 64 + i * 16 + rl;
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    const int colp = colw + jh * 64;
                    float v[4][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const half4v b4 = *(const half4v*)(bias_lds + jh * 64 + j * 16);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[j][r] = r16(acc[i][jh * 4 + j][r] + (float)b4[r]);      // the Linear's fp16 output
                    }
                    if (ACT == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 4; r += 2) {             // two values per packed fp32 instruction, same arithmetic
                                const float2v y = gelu_erf2(float2v{v[j][r], v[j][r + 1]});
                                v[j][r] = r16(y[0]); v[j][r + 1] = r16(y[1]);
                            }
                    } else if (ACT == 2) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[j][r] = r16(gelu_tanh(v[j][r]));
                    }
                    if (scale_cols) {                                           // q, k * d^-0.25 (torch_model.py:93-95)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float sc = (colp + j * 16 < p.colscale_n) ? p.colscale : 1.0f;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[j][r] = r16(v[j][r] * sc);
                        }
                    }
                    if constexpr (RES) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[j][r] += (float)r4[i][jh * 4 + j][r];
                    }
                    if constexpr (SIMPLE) {
                        if (row < p.M) {
                            h16* crow = p.C + (size_t)row * p.ldc + colp;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                *(half4v*)(crow + j * 16) = half4v{(h16)v[j][0], (h16)v[j][1], (h16)v[j][2], (h16)v[j][3]};
                        }
                    } else if (row < p.M) {
                        if (p.out_mode == 0) {
                            h16* crow = p.C + (p.c_rows > 0 ? (size_t)(row / p.c_rows) * p.c_bstride + (size_t)(row % p.c_rows) * p.ldc
                                                            : (size_t)row * p.ldc) + colp;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                *(half4v*)(crow + j * 16) = half4v{(h16)v[j][0], (h16)v[j][1], (h16)v[j][2], (h16)v[j][3]};
                        } else {       // head-split [B, 2, H, T, 64] (whisper/model.py:519); a lane's 4 channels stay inside one head
                            // (no per-element integer division at Whisper's sizes: the tile's first row is divided once, on the
                            // scalar unit, a tile spans fewer rows than an utterance has positions, and K | V is one comparison;
                            // the divisions cost this epilogue several of a K = 1280 tile's 40 stages)
                            const int HC = p.hs_H * 64;
                            int bb = hs_b0, tt = hs_t0 + (row - row0);
                            if (p.hs_T >= BM) { if (tt >= p.hs_T) { tt -= p.hs_T; ++bb; } }       // at most one utterance boundary inside a tile
                            else { bb = row / p.hs_T; tt = row - bb * p.hs_T; }                   // (tiny models: several utterances per tile)
                            const bool two = p.N == 2 * HC;                                        // K | V side by side: one comparison
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int col = colp + j * 16;
                                const int kv = p.hs_kv < 0 ? (two ? (col >= HC ? 1 : 0) : col / HC) : p.hs_kv, cc = p.hs_kv < 0 ? col - kv * HC : col;
                                const size_t off = ((((size_t)bb * 2 + kv) * p.hs_H + (cc >> 6)) * p.hs_T + tt) * 64 + (cc & 63);
                                if (p.q8_inv_scale > 0.f) {      // int8 cross K/V (opt-in): the fp16 result, quantised like the self-attention cache
                                    char4 q;
                                    q.x = (signed char)fminf(127.f, fmaxf(-128.f, rintf(r16(v[j][0]) * p.q8_inv_scale)));
                                    q.y = (signed char)fminf(127.f, fmaxf(-128.f, rintf(r16(v[j][1]) * p.q8_inv_scale)));
                                    q.z = (signed char)fminf(127.f, fmaxf(-128.f, rintf(r16(v[j][2]) * p.q8_inv_scale)));
                                    q.w = (signed char)fminf(127.f, fmaxf(-128.f, rintf(r16(v[j][3]) * p.q8_inv_scale)));
                                    *(char4*)((signed char*)p.C + off) = q;
                                } else {
                                    *(half4v*)(p.C + off) = half4v{(h16)v[j][0], (h16)v[j][1], (h16)v[j][2], (h16)v[j][3]};
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // SIMPLE (plain row-major output and residual: the encoder layers' four GEMMs), round 3: the same values with a third of the
        // instructions.  The general form above converts this lane's bias values once per 16-row block and sends every output
        // through fp32 -> fp16 -> fp32 -> fp16; it compiled to ~1000 instructions per wave and tile -- 17 000 cycles of a plain
        // K = 1280 tile's 84 000 with two waves per SIMD in it together, 32 000 with a residual (from the shapes' rates: 160 S + E
        // and 40 S + E cycles per tile at K = 5120 / 1280).  Here: bias to fp32 ONCE per tile (32 values per lane), packed adds,
        // one conversion per output where nothing sits between the Linear's rounding and the store.
        auto finish_simple = [&](auto res_tag, auto scale_tag) {
            constexpr bool RES = decltype(res_tag)::value, SCALE = decltype(scale_tag)::value;
            // 16-BYTE accesses.  In the accumulator layout a lane owns 4 consecutive channels of a row (8 bytes) per 16-channel block,
            // the lane 16 places on the next 4: every store instruction wrote 16 rows x 32 bytes, and the CU's store path, which
            // works segment by segment, needed 3.5 us for a wave's 32 of them (time stamps inside the kernel: a ~600-instruction
            // epilogue took 4.8 us).  Two adjacent blocks are therefore exchanged between the paired lane rows first
            // (v_permlane16_swap: lanes of rows 0 / 2 end up with 8 consecutive channels of block 2 jp, rows 1 / 3 with those of
            // block 2 jp + 1): 16 stores of 16 rows x 64 contiguous bytes, and the residual arrives by 16-byte loads in the same
            // layout.  Element by element the same arithmetic as before.
            const int odd = ge & 1;
            const int colx = col0 + wc * 128 + 4 * (ge - odd) + 16 * odd;          // + 32 jp: this lane's first channel after the exchange
            uint4 r8[RES ? 4 : 1][RES ? 4 : 1];
            if constexpr (RES) {
#pragma unroll
                for (int i = 0; i < (F16R_FULL_LINE_EPI ? 0 : (F16R_EPI_SPLIT ? 2 : 4)); ++i) {
                    const int row = row0 + wr * 64 + i * 16 + rl;
                    const h16* rrow = p.residual + (size_t)(row < p.M ? row : p.M - 1) * p.ldr + colx;
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        if (i < F16R_RES_EARLY) r8[i][jp] = r8e[i][jp];      // (requested while the last slot was multiplied)
                        else if (F16R_NT_EPI & 2) r8[i][jp] = __builtin_bit_cast(uint4, __builtin_nontemporal_load((const u32x4*)(rrow + jp * 32)));
                        else r8[i][jp] = *(const uint4*)(rrow + jp * 32);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float2v bf[8][2];                                 // bias of this lane's channels colw + 16 j + (0..3), as fp32 pairs
            float scj[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const half4v b4 = *(const half4v*)(bias_lds + j * 16);
                bf[j][0] = float2v{(float)b4[0], (float)b4[1]};
                bf[j][1] = float2v{(float)b4[2], (float)b4[3]};
                scj[j] = (SCALE && colw + j * 16 < p.colscale_n) ? p.colscale : 1.0f;
            }
            // block j of row block i, everything in front of the residual add, as two packed fp16 pairs (accumulator layout)
            auto block = [&](int i, int j, uint32_t& lo, uint32_t& hi) {
                float2v v0 = float2v{acc[i][j][0], acc[i][j][1]} + bf[j][0];
                float2v v1 = float2v{acc[i][j][2], acc[i][j][3]} + bf[j][1];
                half2v h0 = __builtin_convertvector(v0, half2v), h1 = __builtin_convertvector(v1, half2v);     // the Linear's fp16 output
                if constexpr (ACT != 0 || SCALE) {
                    v0 = __builtin_convertvector(h0, float2v); v1 = __builtin_convertvector(h1, float2v);
                    if (ACT == 1) {
                        v0 = gelu_erf2(v0); v1 = gelu_erf2(v1);
                        if constexpr (SCALE) {
                            h0 = __builtin_convertvector(v0, half2v); h1 = __builtin_convertvector(v1, half2v);
                            v0 = __builtin_convertvector(h0, float2v); v1 = __builtin_convertvector(h1, float2v);
                        }
                    } else if (ACT == 2) {
                        v0 = float2v{gelu_tanh(v0[0]), gelu_tanh(v0[1])};
                        v1 = float2v{gelu_tanh(v1[0]), gelu_tanh(v1[1])};
                        if constexpr (SCALE) {
                            h0 = __builtin_convertvector(v0, half2v); h1 = __builtin_convertvector(v1, half2v);
                            v0 = __builtin_convertvector(h0, float2v); v1 = __builtin_convertvector(h1, float2v);
                        }
                    }
                    if constexpr (SCALE) {                    // q, k * d^-0.25 (torch_model.py:93-95)
                        v0 = float2v{v0[0] * scj[j], v0[1] * scj[j]};
                        v1 = float2v{v1[0] * scj[j], v1[1] * scj[j]};
                    }
                    h0 = __builtin_convertvector(v0, half2v); h1 = __builtin_convertvector(v1, half2v);
                }
                lo = __builtin_bit_cast(uint32_t, h0); hi = __builtin_bit_cast(uint32_t, h1);
            };
            if constexpr (F16R_FULL_LINE_EPI != 0) {
                // lane (rl, ge) holds, of row rl of a row block, the 16-byte chunks A (channel block pair 2 m) and B (2 m + 1).  Lanes rl and rl ^ 8
                // exchange one of them, so that instruction 1 carries rows 0-7 (lanes rl < 8: their A = the first 64 bytes of the row's 128, lanes
                // rl >= 8: B of row rl - 8 = the second 64) and instruction 2 rows 8-15 (lanes rl >= 8: their B, lanes rl < 8: A of row rl + 8).
                const int lo8 = rl < 8;
                const int colq = col0 + wc * 128 + 4 * (ge - odd) + 16 * odd + (rl >> 3) * 32;      // + 64 m: first channel of this lane's chunk in either instruction
                auto ror8 = [&](uint4 x) {
                    uint4 y;
                    y.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.x, 0x128, 0xf, 0xf, false);
                    y.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.y, 0x128, 0xf, 0xf, false);
                    y.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.z, 0x128, 0xf, 0xf, false);
                    y.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.w, 0x128, 0xf, 0xf, false);
                    return y;
                };
                auto add_res = [&](uint4 o, uint4 rr) {
                    const half8v x = __builtin_bit_cast(half8v, o), r = __builtin_bit_cast(half8v, rr);
                    half8v y;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float2v tt = float2v{(float)x[e], (float)x[e + 1]} + float2v{(float)r[e], (float)r[e + 1]};
                        const half2v h = __builtin_convertvector(tt, half2v);
                        y[e] = h[0]; y[e + 1] = h[1];
                    }
                    return __builtin_bit_cast(uint4, y);
                };
                uint4 rq[RES ? 4 : 1][RES ? 2 : 1][RES ? 2 : 1];      // [row block][m][instruction]
                if constexpr (RES) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int row = row0 + wr * 64 + i * 16 + 8 * k + (rl & 7);
                            const h16* rrow = p.residual + (size_t)(row < p.M ? row : p.M - 1) * p.ldr + colq;
#pragma unroll
                            for (int m = 0; m < 2; ++m) rq[i][m][k] = *(const uint4*)(rrow + m * 64);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        uint32_t a0, a1, b0, b1, c0, c1, d0, d1;
                        block(i, 4 * m, a0, a1);
                        block(i, 4 * m + 1, b0, b1);
                        lane_rows_swap16(a0, b0);
                        lane_rows_swap16(a1, b1);
                        block(i, 4 * m + 2, c0, c1);
                        block(i, 4 * m + 3, d0, d1);
                        lane_rows_swap16(c0, d0);
                        lane_rows_swap16(c1, d1);
                        const uint4 A = make_uint4(a0, a1, b0, b1), B = make_uint4(c0, c1, d0, d1);      // channel block pairs 2 m | 2 m + 1 of row rl
                        const uint4 give = lo8 ? B : A;
                        const uint4 got = ror8(give);                                                        // lane rl ^ 8's
                        uint4 o1 = lo8 ? A : got, o2 = lo8 ? got : B;
                        if constexpr (RES) { o1 = add_res(o1, rq[i][m][0]); o2 = add_res(o2, rq[i][m][1]); }
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int row = row0 + wr * 64 + i * 16 + 8 * k + (rl & 7);
                            h16* crow = p.C + (size_t)(row < p.M ? row : p.M - 1) * p.ldc + colq;
                            if (row < p.M) *(uint4*)(crow + m * 64) = k ? o2 : o1;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
            if constexpr (RES && F16R_EPI_SPLIT) {
                // (r8 holds row blocks 0 and 1 only here: see the loads above)
                uint4 ob[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        uint32_t a0, a1, b0, b1;
                        block(i, 2 * jp, a0, a1);
                        block(i, 2 * jp + 1, b0, b1);
                        lane_rows_swap16(a0, b0);
                        lane_rows_swap16(a1, b1);
                        ob[i][jp] = make_uint4(a0, a1, b0, b1);
                    }
                __builtin_amdgcn_sched_barrier(0);
                uint4 r8b[2][4];
#pragma unroll
                for (int i = 2; i < 4; ++i) {
                    const int row = row0 + wr * 64 + i * 16 + rl;
                    const h16* rrow = p.residual + (size_t)(row < p.M ? row : p.M - 1) * p.ldr + colx;
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) r8b[i - 2][jp] = *(const uint4*)(rrow + jp * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = row0 + wr * 64 + i * 16 + rl;
                    h16* crow = p.C + (size_t)(row < p.M ? row : p.M - 1) * p.ldc + colx;
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        const half8v x = __builtin_bit_cast(half8v, ob[i][jp]), r = __builtin_bit_cast(half8v, i < 2 ? r8[i][jp] : r8b[i - 2][jp]);
                        half8v y;
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const float2v tt = float2v{(float)x[e], (float)x[e + 1]} + float2v{(float)r[e], (float)r[e + 1]};
                            const half2v h = __builtin_convertvector(tt, half2v);
                            y[e] = h[0]; y[e + 1] = h[1];
                        }
                        if (row < p.M) *(uint4*)(crow + jp * 32) = __builtin_bit_cast(uint4, y);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + wr * 64 + i * 16 + rl;
                h16* crow = p.C + (size_t)(row < p.M ? row : p.M - 1) * p.ldc + colx;
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    uint32_t a0, a1, b0, b1;
                    block(i, 2 * jp, a0, a1);
                    block(i, 2 * jp + 1, b0, b1);
                    lane_rows_swap16(a0, b0);                 // even lane rows: (a, b) = block 2 jp, own 4 channels | the next 4;
                    lane_rows_swap16(a1, b1);                 // odd lane rows: block 2 jp + 1, the previous 4 | own
                    uint4 o = make_uint4(a0, a1, b0, b1);
                    if constexpr (RES) {
                        const half8v x = __builtin_bit_cast(half8v, o), r = __builtin_bit_cast(half8v, r8[i][jp]);
                        half8v y;
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const float2v t = float2v{(float)x[e], (float)x[e + 1]} + float2v{(float)r[e], (float)r[e + 1]};
                            const half2v h = __builtin_convertvector(t, half2v);
                            y[e] = h[0]; y[e + 1] = h[1];
                        }
                        o = __builtin_bit_cast(uint4, y);
                    }
                    if (row < p.M) {
                        if (F16R_NT_EPI & 1) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), (u32x4*)(crow + jp * 32));
                        else *(uint4*)(crow + jp * 32) = o;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (SIMPLE && ACT != 0) {
            finish_simple(std::false_type{}, std::false_type{});                         // (the launcher sends GELU + residual / column scale to the general form)
        } else if constexpr (SIMPLE) {     // (wave-uniform branches, once per tile)
            if (p.residual) finish_simple(std::true_type{}, std::false_type{});          // no encoder GEMM has both a residual and a column scale
            else if (scale_cols) finish_simple(std::false_type{}, std::true_type{});
            else finish_simple(std::false_type{}, std::false_type{});
        } else {
            if (p.residual) finish(std::true_type{}); else finish(std::false_type{});
        }
        // the store count the next stage wait adds is exact only for a tile without an M tail (rows past M skip their stores)
        if (F16R_EPI_ISSUE) { if (row0 + BM <= p.M) after_epilogue = true; else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (++t == my_tiles) break;
        zero_acc();
        if (wid >= 4) __builtin_amdgcn_s_barrier();  // one barrier behind waves 0-3 again
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the loader's run-ahead requests (never read) land before the workgroup's LDS is released
}

bool gemm_f16r_supports(const GemmBigParams& p) {
    return p.N % f16r::BN == 0 && p.N <= f16r::MAX_N && p.K % 64 == 0 && p.K >= 128;
}

int launch_gemm_f16r(const GemmBigParams& p, hipStream_t stream) {
    using namespace f16r;
    WM_REQUIRE(p.N % BN == 0 && p.N <= MAX_N, "gemm_f16r: N=%d must be a multiple of %d, <= %d", p.N, BN, MAX_N);
    WM_REQUIRE(p.K % 64 == 0 && p.K >= 128, "gemm_f16r: K=%d must be a multiple of 64, >= 128", p.K);
    WM_REQUIRE(p.lda % 8 == 0, "gemm_f16r: lda=%d must be a multiple of 8 (16-byte loads)", p.lda);
    WM_REQUIRE(p.ldc % 4 == 0 && p.ldr % 4 == 0, "gemm_f16r: ldc/ldr must be multiples of 4 (8-byte epilogue accesses)");
    WM_REQUIRE(p.M > 0, "gemm_f16r: empty M");
    WM_REQUIRE(p.act >= 0 && p.act <= 2, "gemm_f16r: act=%d", p.act);
    constexpr int STAGES = 4;                        // 128 KB ring + 16 KB bias of the CU's 160 KB (5 stages measured no faster)
    static std::atomic<int> n_cu_dev[64];
    int dev = 0;
    WM_CHECK_HIP(hipGetDevice(&dev));
    const int slot = (dev >= 0 && dev < 64) ? dev : 0;
    int n_cu = n_cu_dev[slot].load(std::memory_order_relaxed);
    using Kern = void (*)(GemmBigParams);
    static const Kern kerns[6] = {gemm_f16r_kernel<STAGES, 0>, gemm_f16r_kernel<STAGES, 1>, gemm_f16r_kernel<STAGES, 2>,
                                  gemm_f16r_kernel<STAGES, 0, true>, gemm_f16r_kernel<STAGES, 1, true>, gemm_f16r_kernel<STAGES, 2, true>};
    constexpr size_t LDS_BYTES = (size_t)2 * SLOT + MAX_N * 2;
    if (n_cu == 0) {
        int v = 0;
        WM_CHECK_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        n_cu = v > 0 ? v : 256;
        for (int a = 0; a < 6; ++a)
            WM_CHECK_HIP(hipFuncSetAttribute((const void*)kerns[a], hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        n_cu_dev[slot].store(n_cu, std::memory_order_relaxed);
    }
    // one workgroup per CU (a workgroup holds 128 of the CU's 160 KB of LDS), a multiple of 8 so that every XCD gets the same count
    const int n_tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    int grid = (n_cu / 8) * 8;
    if (grid < 8) grid = 8;
    static const int lab_wgs = 0;      // probes only (scripts/kv_beside_probe.py; honoured under WM_LAB=1)
    const int max_wgs = p.max_wgs > 0 ? p.max_wgs : lab_wgs;
    if (max_wgs > 0 && max_wgs < grid) grid = max_wgs >= 8 ? (max_wgs / 8) * 8 : 8;
    const int need = ((n_tiles + 7) / 8) * 8;        // never more workgroups than a band has tiles
    if (grid > need) grid = need;
    const bool simple = p.out_mode == 0 && p.c_rows == 0 && p.res_mod == 0 && p.q8_inv_scale <= 0.f && !(p.residual && p.colscale_n > 0) &&
                        (p.act == 0 || (!p.residual && p.colscale_n <= 0)) &&
                        p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 && (!p.residual || (p.ldr % 8 == 0 && ((uintptr_t)p.residual & 15) == 0));      // 16-byte epilogue accesses
    static const int lab_rows = 0;    // A/B runs (WM_LAB=1): 1 = the plain row-major tile order
    GemmBigParams q = p;
    if (q.tile_rows <= 0) q.tile_rows = lab_rows;
    hipLaunchKernelGGL(kerns[p.act + (simple ? 3 : 0)], dim3(grid), dim3(512), LDS_BYTES, stream, q);
    WM_LAUNCH_CHECK(stream, "gemm_f16r");
    return 0;
}

}  // namespace wm
