#!/bin/bash
set -e
B=tools/scan_exp/build_variant.sh
NS='rep("            if (!STATE_ONLY) finish(k - 1, k0);\n            park(k + 1, k0);\n            fetch(k + 2);\n", ""); rep("                if (!STATE_ONLY) finish(k, k1);\n                park(k + 2, k1);\n                fetch(k + 3);\n", "");'
DUAL='rep("template <typename T, typename TBC, bool STATE_ONLY>\n__global__ __launch_bounds__(512) void sscan2_fwd_kernel(const S2Fwd p) {\n    typedef Row4<T> R;\n    typedef Row4<TBC> RBC;\n    __shared__ __attribute__((aligned(16))) FTile tiles[2];\n    __shared__ __attribute__((aligned(16))) float ypart[STATE_ONLY ? TT * CB : 2 * YP_TILE];   // STATE_ONLY: the staging lanes'"'"' sums of dt on their way to sdelta\n    const bool staging = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;", "template <typename T, typename TBC, bool STATE_ONLY>\n__global__ __launch_bounds__(512) void sscan2_fwd_kernel(const S2Fwd p) {\n    typedef Row4<T> R;\n    typedef Row4<TBC> RBC;\n    __shared__ __attribute__((aligned(16))) FTile tiles[2];\n    __shared__ __attribute__((aligned(16))) float ypart[STATE_ONLY ? TT * CB : 2 * YP_TILE];\n    const bool staging = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 1024;");'
$B s0 "$NS"
$B s_dual "$DUAL"
