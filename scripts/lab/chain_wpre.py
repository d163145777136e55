"""Lab: the one-launch step with the WIDE stage's (mlp2) weights requested into registers from the stage before it (mlp1), right behind mlp1's
multiply, and mlp1's two closing barriers as raw s_barriers (lgkmcnt only: __syncthreads() would wait for the loads in flight).
    python scripts/lab/chain_wpre.py      -> build/lab/gemv_chain_wpre.hip + build/lab/libwm_wpre.so
Round 4's third batch-1 lever in its register form (the LDS-DMA form lost 3 %: profiles/r5af_*).  Nothing here ships."""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "eddie-wang-hackathon2023_amd/csrc/gemv_chain.hip")).read()


def rep(old, new, count=1):
    global s
    assert s.count(old) >= 1, old[:80]
    s = s.replace(old, new, count)


# 1. signature: the prefetched tiles, whether they are there, and what to prefetch for the stage behind
rep('''                                            int in_kind, const unsigned long long* gran, unsigned tag, bool x_in_granules) {
    // in_kind == CHAIN_IN_LDS''', '''                                            int in_kind, const unsigned long long* gran, unsigned tag, bool x_in_granules,
                                            u32x4 (&wpre)[2][WB == 16 ? 10 : 5], bool& have_pre, const void* nxt_wt, int nxt_K, int nxt_nb) {
    // in_kind == CHAIN_IN_LDS''')
# 2. a loader for the wide stage's tiles of this wave (its own code, for the NEXT stage's matrix)
rep('''    // an idle slot (no group of this stage falls to it) only keeps the workgroup's barriers company''', '''    // (lab) the NEXT stage is the wide one: every wave of a workgroup that owns a group of it requests ITS tiles of that stage now / behind the multiply
    const bool pre = !WIDE && nxt_wt != nullptr && (int)blockIdx.x < nxt_nb;      // (workgroup-uniform)
    auto prefetch_wide = [&]() {
        const int ktn = nxt_K / KT, sl = (ktn + TB - 1) / TB, tpsn = (ktn + sl - 1) / sl;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int slice = wid + 8 * j;
            const int tb = min(slice, sl - 1) * tpsn, tl = max(min(ktn, tb + tpsn) - 1, 0);
            const __attribute__((address_space(1))) u32x4* wt = CHAIN_GLOBAL(u32x4, nxt_wt) + (size_t)blockIdx.x * ktn * 64 + lane;
            if (slice < sl) {
#pragma unroll
                for (int i = 0; i < TB; ++i) wpre[j][i] = __builtin_nontemporal_load(wt + (size_t)min(tb + i, tl) * 64);
            } else {
#pragma unroll
                for (int i = 0; i < TB; ++i) wpre[j][i] = u32x4{0u, 0u, 0u, 0u};
            }
        }
        have_pre = true;
    };
    auto barrier = [&]() {                           // behind a prefetch: LDS traffic only is waited for
        if (pre) asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");
        else __syncthreads();
    };
    // an idle slot (no group of this stage falls to it) only keeps the workgroup's barriers company''')
rep('''        const int nbar = LN ? 3 : 2;
        for (int b = 0; b < nbar; ++b) __syncthreads();''', '''        const int nbar = LN ? 3 : 2;
        if (pre) prefetch_wide();
        for (int b = 0; b < nbar; ++b) barrier();''')
# 3. the wide stage takes the prefetched tiles
rep('''        if (slice < slices && has_group) {                    // (wave-uniform: an idle slot or an absent slice streams nothing)
#pragma unroll
            for (int i = 0; i < TB; ++i) wreg[j][i] = __builtin_nontemporal_load(wt + (size_t)min(t_begin[j] + i, t_last) * 64);
        } else {''', '''        if (WIDE && have_pre) {                               // (lab) requested by the stage before
#pragma unroll
            for (int i = 0; i < TB; ++i) wreg[j][i] = wpre[j < 2 ? j : 0][i];
        } else if (slice < slices && has_group) {             // (wave-uniform: an idle slot or an absent slice streams nothing)
#pragma unroll
            for (int i = 0; i < TB; ++i) wreg[j][i] = __builtin_nontemporal_load(wt + (size_t)min(t_begin[j] + i, t_last) * 64);
        } else {''')
# 4. the prefetch behind the multiply, raw barriers behind it (the LayerNorm's barrier lies in front: untouched)
rep('''    __syncthreads();

    // ---- 4. epilogue of the slot''', '''    if (pre) prefetch_wide();
    barrier();

    // ---- 4. epilogue of the slot''')
rep('''    if (st.mode == 2) own_valid = true;
    __syncthreads();                                          // s_red / s_in are the next stage's
}''', '''    if (st.mode == 2) own_valid = true;
    barrier();                                                // s_red / s_in are the next stage's
    if (WIDE) have_pre = false;
}''')
# 5. kernel main: registers, the next stage's matrix
rep('''    bool own_valid = false, x_in_granules = false;
    for (int l = whole ? -1 : 0;''', '''    u32x4 wpre[2][WB == 16 ? 10 : 5];
    bool have_pre = false;
    bool own_valid = false, x_in_granules = false;
    for (int l = whole ? -1 : 0;''')
rep('''            const bool wide = (st.K / KT + TB - 1) / TB > 4;''', '''            const bool wide = (st.K / KT + TB - 1) / TB > 4;
            const void* nxt_wt = nullptr; int nxt_K = 0, nxt_nb = 0;
            if (!wide && s + 1 < s_end) {                                      // (uniform) is the stage behind the wide one?
                const int idx2 = (whole ? 1 + 6 * l + s : s) + 1;
                static_assert(offsetof(ChainStage, Wt) == 0 && offsetof(ChainStage, K) == 40 && offsetof(ChainStage, n_blocks) == 44, "descriptor layout");
                const unsigned w0 = (unsigned)__builtin_amdgcn_readfirstlane((int)s_desc[idx2 * DESC_DW + 0]), w1 = (unsigned)__builtin_amdgcn_readfirstlane((int)s_desc[idx2 * DESC_DW + 1]);
                const int k2 = __builtin_amdgcn_readfirstlane((int)s_desc[idx2 * DESC_DW + 10]), nb2 = __builtin_amdgcn_readfirstlane((int)s_desc[idx2 * DESC_DW + 11]);
                if ((k2 / KT + TB - 1) / TB > 4) { nxt_wt = (const void*)(((unsigned long long)w1 << 32) | w0); nxt_K = k2; nxt_nb = nb2; }
            }''')
for kind in ('true, false', 'false, true', 'false, false'):
    rep(f'chain_stage<WB, {kind}, NR>(p, st, s, epoch, own_valid, s_red, s_in, s_own, in_kind, gran, tag, x_in_granules);',
        f'chain_stage<WB, {kind}, NR>(p, st, s, epoch, own_valid, s_red, s_in, s_own, in_kind, gran, tag, x_in_granules, wpre, have_pre, nxt_wt, nxt_K, nxt_nb);')
out = os.path.join(ROOT, "build/lab")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "gemv_chain_wpre.hip"), "w").write(s)
env = dict(os.environ, SRC=os.path.join(out, "gemv_chain_wpre.hip"))
r = subprocess.run([os.path.join(ROOT, "scripts/lab/build_variant.sh"), "wpre", "gemv_chain.hip", "-Rpass-analysis=kernel-resource-usage"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
import re
names = re.findall(r"Function Name: (\S+)", r.stdout); vg = re.findall(r"VGPRs: (\d+)", r.stdout); sc = re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stdout)
for n, v, c in zip(names, vg, sc):
    if "gemv_chain_kernelILi8E" in n or "error" in n:
        print(n[20:60], "VGPRs", v, "scratch", c)
print([l for l in r.stdout.splitlines() if "error" in l][:5])
print(r.stdout.strip().splitlines()[-1])
