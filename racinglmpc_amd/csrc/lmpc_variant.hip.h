// racinglmpc_amd/csrc/lmpc_variant.hip.h -- one (N, numSS_points) instantiation of the solve kernels behind a small table of launchers.
//
// The solve kernels are templates on the horizon N and the number of safe-set columns S (every LDS offset, trip count and register array
// is a compile-time constant).  The library carries the reference's configurations built in (lmpc_capi.hip: N in {8,12,14,20,40} x S in
// {0,48}); any other pair -- the reference takes any N (main.py:43) and numSS_Points = 12 numSS_it (initControllerParameters.py:43-44) -- is
// a shared object of its own, liblmpc_var_N<N>_S<S>.so next to the library (racinglmpc_amd/csrc/lmpc_variant.hip compiled with
// -DLMPC_VAR_N / -DLMPC_VAR_S; racinglmpc_amd.build.build_variant), loaded by lmpc_create on demand.
#pragma once
#include "lmpc_kernels.hip.h"
#include "lmpc_solve_mw.hip.h"
#ifndef LMPC_VARIANT_TU
#include "lmpc_solve_rt.hip.h"       // the runtime-(N, S) fallback kernel lives in the library only
#endif
#ifdef LMPC_WITH_CD                  // the condensed kernel is a measured alternative (not faster), kept out of the default build: racinglmpc_amd.build.build_flavour("cd", ["LMPC_WITH_CD"])
#include "lmpc_solve_cd.hip.h"
#endif

struct lmpc_variant_api {
    int N, S;
    size_t lds_mw, lds_1w;                    // dynamic LDS per QP: multi-wave kernels / one-wave kernels
    int (*launch_1w)(hipStream_t, const lmpc_dev_params &, int B, const lmpc_solve_io &);      // lmpc_solve_kernel<N,S>
    int (*launch_retry)(hipStream_t, const lmpc_dev_params &, int B, const lmpc_solve_io &);   // lmpc_solve_kernel<N,S,true>
    int (*launch_mw4)(hipStream_t, const lmpc_dev_params &, int B, const lmpc_solve_io &);     // lmpc_solve_kernel_mw<N,S,4>
    int (*launch_mw2)(hipStream_t, const lmpc_dev_params &, int B, const lmpc_solve_io &);     // lmpc_solve_kernel_mw<N,S,2>
    size_t lds_1w_abg;                        // one-wave kernel with [A_k | B_k] in global memory (long horizons): dynamic LDS per QP; 0 = not used for this (N, S)
    size_t lds_cd;                            // condensed one-wave kernel (lmpc_solve_cd.hip.h): dynamic LDS per QP without the state-cost block; 0 = not built for this (N, S)
    size_t lds_cd_q;                          //   ... with it (Q or Qf non-zero)
    int (*launch_cd)(hipStream_t, const lmpc_dev_params &, int B, const lmpc_solve_io &, int hasQ);   // lmpc_solve_kernel_cd<N,S>
    int occ_mw2;                              // work-groups of the two-wave kernel the runtime keeps resident per CU (asked, not assumed: 0 = unknown)
};

// What lmpc_create asks a variant library for (lmpc_variant_get): the revision and the sizes of the three structures that cross the boundary by value or
// by layout.  A liblmpc_var_*.so built against another layout of the parameter block answers -1 and is rebuilt (racinglmpc_amd._capi: E_VARIANT).
constexpr int LMPC_VARIANT_ABI = LMPC_VARIANT_ABI_REV * 0x1000000 + (int)((sizeof(lmpc_dev_params) * 31u + sizeof(lmpc_solve_io) * 7u + sizeof(lmpc_variant_api)) & 0xffffffu);
static_assert(sizeof(lmpc_dev_params) < 4096 && sizeof(lmpc_solve_io) < 1024, "by-value kernel arguments: keep them inside the kernarg segment's preload range");

template <int N, int S> struct lmpc_variant_launchers {
    static constexpr bool has_mw = true;               // (every supported S: wave 0 of the multi-wave kernels carries ceil((S + 6) / 64) terminal-block columns per lane)
    static constexpr size_t lds1 = (size_t)solve_lds1<N, S>::tot * sizeof(double), ldsm = has_mw ? (size_t)solve_lds<N, S>::tot * sizeof(double) : 0;
    // fused step (io.mode & 4): the regression's work space sits behind [A_k | B_k], C_k; it fits the solve's footprint at the reference's
    // settings (4 laps x 7 points) and grows it a little beyond
    static size_t lds_for(const lmpc_dev_params &p, const lmpc_solve_io &io) {
        const size_t f = (io.mode & 4) ? (size_t)(54 * N + k1_fused_doubles(N, p.trToUse, p.maxNumPoint)) * sizeof(double) : 0;
        return f > lds1 ? f : lds1;
    }
    static constexpr size_t lds1_max() { const size_t f = (size_t)(54 * N + k1_fused_doubles(N, LMPC_MAX_USED_LAPS, 8)) * sizeof(double); return f > lds1 ? f : lds1; }
    // long horizons: [A_k | B_k] in global memory where that lets more QPs share a CU than the 160 KB of LDS otherwise hold and fewer than four do (idle SIMDs)
    static constexpr size_t lds1g = (size_t)solve_lds1<N, S, true>::tot * sizeof(double);
    // (measured, solve kernel, ms at batch 1024 / 4096: N = 40 LDS 1.467 / 4.073, global 1.092 / 3.398 -- two -> four QPs per CU;  N = 20 LDS 0.655 / 1.571,
    //  global 0.725 / 1.554 -- five -> seven per CU, but every SIMD was busy already and the variant spills: only where fewer than four QPs fit)
    static constexpr bool use_abg = (160 * 1024 / lds1) < 4 && (160 * 1024 / lds1g) > (160 * 1024 / lds1) && solve_lds1<N, S>::CH == 1;
    static int l1(hipStream_t st, const lmpc_dev_params &p, int B, const lmpc_solve_io &io) {
        if constexpr (use_abg) {
            if (!(io.mode & 4) && io.abPack) { hipLaunchKernelGGL((lmpc_solve_kernel<N, S, false, true>), dim3(B), dim3(WAVE), lds1g, st, p, B, io); return 0; }
        }
        hipLaunchKernelGGL((lmpc_solve_kernel<N, S>), dim3(B), dim3(WAVE), lds_for(p, io), st, p, B, io); return 0; }
    static int lr(hipStream_t st, const lmpc_dev_params &p, int B, const lmpc_solve_io &io) {
        hipLaunchKernelGGL((lmpc_solve_kernel<N, S, true>), dim3(B), dim3(WAVE), lds_for(p, io), st, p, B, io); return 0; }
    static int l4(hipStream_t st, const lmpc_dev_params &p, int B, const lmpc_solve_io &io) {
        if constexpr (has_mw) { hipLaunchKernelGGL((lmpc_solve_kernel_mw<N, S, 4>), dim3(B), dim3(WAVE * 4), ldsm, st, p, B, io); return 0; } else return l1(st, p, B, io); }
#ifdef LMPC_WITH_CD
    static constexpr bool has_cd = 2 * N <= 32 && S + 6 <= WAVE;          // condensed kernel: short horizons, one terminal-block column per lane
#else
    static constexpr bool has_cd = false;
#endif
    static int lc(hipStream_t st, const lmpc_dev_params &p, int B, const lmpc_solve_io &io, int hasQ) {
#ifdef LMPC_WITH_CD
        if constexpr (has_cd) {
            const size_t lds = (size_t)(hasQ ? solve_ldsc<N, S>::tot_q : solve_ldsc<N, S>::tot) * sizeof(double);
            hipLaunchKernelGGL((lmpc_solve_kernel_cd<N, S>), dim3(B), dim3(WAVE), lds, st, p, B, io, hasQ); return 0;
        }
#endif
        (void)hasQ; return l1(st, p, B, io);
    }
    static int l2(hipStream_t st, const lmpc_dev_params &p, int B, const lmpc_solve_io &io) {
        if constexpr (has_mw) { hipLaunchKernelGGL((lmpc_solve_kernel_mw<N, S, 2>), dim3(B), dim3(WAVE * 2), ldsm, st, p, B, io); return 0; } else return l1(st, p, B, io); }
};

// fills the table and raises the kernels' dynamic-LDS limit; false if the device refuses (footprint beyond 160 KB)
template <int N, int S> static bool lmpc_variant_fill(lmpc_variant_api *v) {
    static_assert(N >= 2 && N <= LMPC_MAX_N && S >= 0 && S <= LMPC_MAX_SS_POINTS, "unsupported variant");
    using L = lmpc_variant_launchers<N, S>;
    v->N = N; v->S = S;
    v->lds_mw = L::ldsm; v->lds_1w = (size_t)solve_lds1<N, S>::tot * sizeof(double);      // lds_mw == 0: no multi-wave kernels for this S
    if (hipFuncSetAttribute((const void *)lmpc_solve_kernel<N, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::lds1_max()) != hipSuccess) return false;
    if (hipFuncSetAttribute((const void *)lmpc_solve_kernel<N, S, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::lds1_max()) != hipSuccess) return false;
    if constexpr (L::has_mw) {
        if (hipFuncSetAttribute((const void *)lmpc_solve_kernel_mw<N, S, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->lds_mw) != hipSuccess) return false;
        if (hipFuncSetAttribute((const void *)lmpc_solve_kernel_mw<N, S, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->lds_mw) != hipSuccess) return false;
    }
    v->launch_1w = &L::l1; v->launch_retry = &L::lr; v->launch_mw4 = &L::l4; v->launch_mw2 = &L::l2;
    v->occ_mw2 = 0;
    if constexpr (L::has_mw) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)lmpc_solve_kernel_mw<N, S, 2>, 2 * WAVE, v->lds_mw) == hipSuccess) v->occ_mw2 = nb;
    }
    v->lds_1w_abg = 0;
    if constexpr (L::use_abg) {
        v->lds_1w_abg = L::lds1g;
        if (hipFuncSetAttribute((const void *)lmpc_solve_kernel<N, S, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::lds1g) != hipSuccess) return false;
    }
    v->lds_cd = 0; v->lds_cd_q = 0; v->launch_cd = &L::lc;
#ifdef LMPC_WITH_CD
    if constexpr (L::has_cd) {
        v->lds_cd = (size_t)solve_ldsc<N, S>::tot * sizeof(double); v->lds_cd_q = (size_t)solve_ldsc<N, S>::tot_q * sizeof(double);
        if (hipFuncSetAttribute((const void *)lmpc_solve_kernel_cd<N, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->lds_cd_q) != hipSuccess) return false;
    }
#endif
    return true;
}

#ifndef LMPC_VARIANT_TU
// The runtime-(N, S) kernel behind the same table (lmpc_solve_rt.hip.h): one wave per QP whatever the batch size, no multi-wave / condensed / fused forms.
// false if the footprint of this (N, S) exceeds the LDS of a CU (N = 64 with 384 safe-set points does: 171 KB).
static bool lmpc_variant_fill_rt(lmpc_variant_api *v, int N, int S) {
    if (N < 2 || N > LMPC_MAX_N || S < 0 || S > LMPC_MAX_SS_POINTS) return false;
    const size_t lds = (size_t)rt_layout(N, S).tot * sizeof(double);
    if (lds + 1024 > (size_t)160 * 1024) return false;
    if (hipFuncSetAttribute((const void *)lmpc_solve_kernel_rt<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    if (hipFuncSetAttribute((const void *)lmpc_solve_kernel_rt<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    v->N = N; v->S = S; v->lds_mw = 0; v->lds_1w = lds; v->lds_1w_abg = 0; v->lds_cd = 0; v->lds_cd_q = 0; v->occ_mw2 = 0;
    v->launch_1w = &lmpc_rt_launch; v->launch_retry = &lmpc_rt_launch_retry; v->launch_mw4 = &lmpc_rt_launch; v->launch_mw2 = &lmpc_rt_launch; v->launch_cd = &lmpc_rt_launch_cd;
    return true;
}
#endif
