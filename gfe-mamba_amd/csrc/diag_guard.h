// Diagnostic-build guard.
// The kernels carry timing / ablation switches -- GFE_EXP_* (conv3d.hip), CONVT_EXP_* (convt3d.hip) -- that compile WRONG-RESULT variants
// (no MFMA, no staging, no epilogue ...), an A/B switch that restores a superseded kernel (GFE_ATTN_ALWAYS_TRACK) and in-kernel stamp switches (GFE_S2_STAMPS, GFE_GDMA_STAMPS, GFE_EXP_STAMP) that add
// instrumentation.  They exist for diagnostic libraries only: tools/build_exp.sh and tools/scan_exp/build_variant.sh pass -DGFE_DIAG and write
// to exp_build/, never to gfe_hip/libgfe_hip.so.  Any of them WITHOUT GFE_DIAG is a build error here, csrc/Makefile refuses flags that name
// them, and a diagnostic library exports `gfe_diag_build` (lib.hip), which tests/test_build_resources.py asserts the product library lacks.
// tests/test_build_resources.py also checks that every such switch used in csrc/ is listed below.
#pragma once
#if defined(GFE_EXP_A_AFTER) || defined(GFE_EXP_DMA_FIRST) || defined(GFE_EXP_HALFLDS) || defined(GFE_EXP_KPRIO) || defined(GFE_EXP_NOA) || \
    defined(GFE_EXP_NOBAR) || defined(GFE_EXP_NOEPI) || defined(GFE_EXP_NOMFMA) || defined(GFE_EXP_NOPIPE_OCT) || defined(GFE_EXP_NOSTART) || \
    defined(GFE_EXP_NOW) || defined(GFE_EXP_NO_APRIO) || defined(GFE_EXP_NT_STORE) || defined(GFE_EXP_STAMP) || defined(GFE_EXP_WPRIO) || \
    defined(CONVT_EXP_NOA) || defined(CONVT_EXP_NOEPI) || defined(CONVT_EXP_NORES) || defined(CONVT_EXP_NOSTORE) || defined(CONVT_EXP_NOW) || \
    defined(GFE_S2_STAMPS) || defined(GFE_GDMA_STAMPS) || defined(GFE_ATTN_ALWAYS_TRACK)
#define GFE_DIAG_SWITCH_ACTIVE 1
#if !defined(GFE_DIAG)
#error "GFE_EXP_* / CONVT_EXP_* / *_STAMP* switches build diagnostic libraries only: add -DGFE_DIAG (tools/build_exp.sh does) and never link the result as gfe_hip/libgfe_hip.so"
#endif
#endif
