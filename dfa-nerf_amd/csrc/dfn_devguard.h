// dfn_devguard.h - the developer switches of the kernels cannot reach a product build by accident.
//
// The sources carry measurement switches (ablations that price one piece of a kernel by removing it - they render WRONG
// results at full speed - cycle-counter builds, per-workgroup traces).  Every one of them is listed here, and defining any of
// them without DFN_DEV_BUILD is a compile error; DFN_DEV_BUILD marks the library (dfn_version() ends in " DEV"), and
// dfanerf._lib refuses to load a marked library from the in-tree path (only through DFN_LIB=..., the developer override
// tools/build_variant.sh builds for).  dfa-nerf_amd/build.sh refuses DFN_EXTRA_FLAGS without DFN_DEV_BUILD=1 and then passes
// -DDFN_DEV_BUILD itself; tests/test_pack_plan.py checks all three.
#pragma once
#if defined(DFN_EXP_VMCNT) || defined(DFN_EXP_DBLLDS) || defined(DFN_EXP_NOEPI) || defined(DFN_EXP_NOINIT) ||                \
    defined(DFN_EXP_CONSTMASK) || defined(DFN_EXP_CLAMPCVT) || defined(DFN_REC8_NOAMAX) || defined(DFN_REC8_NOSTORE) ||       \
    defined(DFN_REC_NOMASK) || defined(DFN_REC_NOMASKSTORE) || defined(DFN_PUT_SMALL) || defined(DFN_PUT_EIGHTH) ||           \
    defined(DFN_NOMASK) || defined(DFN_NOPUT) || defined(DFN_WL_NOLDS) || defined(DFN_WL_NOMFMA) || defined(DFN_WL_TRACE) ||  \
    defined(DFN_TIMING) || defined(DFN_PRIO_YOUNG) || defined(DFN_REC32_NOSTORE)
#ifndef DFN_DEV_BUILD
#error "a developer / timing switch (DFN_EXP_*, DFN_*_NO*, DFN_TIMING, DFN_WL_TRACE, ...) is defined without DFN_DEV_BUILD: such a build computes WRONG results or carries instrumentation - build it with DFN_DEV_BUILD=1 (build.sh) or tools/build_variant.sh, never as the product library"
#endif
#endif
