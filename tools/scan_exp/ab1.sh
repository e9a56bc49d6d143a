#!/bin/bash
# ablation set 1 (round 6): where the time of the new scan kernels goes.  Builds here (CPU), run on the GPU box with tools/scan_exp/run_ab.sh
set -e
B=tools/scan_exp/build_variant.sh
$B base 'pass'
$B f_nomath 'rep("float v = softplus_nb(raw);", "float v = raw;"); rep("K.gate[j] = has_z ? siluf_(fz[j]) : 1.f;", "K.gate[j] = fz[j];")'
$B f_nofinish 'rep("            for (int q = 1; q < 8; ++q) acc += *reinterpret_cast<const f4*>(yp + q * PS);\n", "            for (int q = 1; q < 1; ++q) acc += *reinterpret_cast<const f4*>(yp + q * PS);\n")'
$B f_noy 'rep("                if (!STATE_ONLY) yp[(4 * u + s) * 8 * PS] = fmaf(h.y, U.bc[s].w, h.x * U.bc[s].z);\n", "")'
$B f_noexp 'rep("                a[s] = f2{fast_exp2(x.x), fast_exp2(x.y)};\n                xb[s] = f2{U.bc[s].x, U.bc[s].y} * dtu;", "                a[s] = x;\n                xb[s] = f2{U.bc[s].x, U.bc[s].y} * dtu;")'
$B f_nostage 'rep("            if (!STATE_ONLY) finish(k - 1, k0);\n            park(k + 1, k0);\n            fetch(k + 2);\n", ""); rep("                if (!STATE_ONLY) finish(k, k1);\n                park(k + 2, k1);\n                fetch(k + 3);\n", "")'
$B f_noscan 'rep("            load_u(ub, u + 1); SB;\n            step_u(ua, u); SB;\n            if (u + 2 < TT / 4) load_u(ua, u + 2);\n            SB;\n            step_u(ub, u + 1); SB;\n", "")'
$B b_nostage 'rep("            drain(K - i, s1);\n", ""); rep("            park(K - 2 - i, stg[(i + 1) & 1], s1);\n            fetch(K - 3 - i);\n", ""); rep("                drain(K - 1 - i, s0);\n", ""); rep("                park(K - 3 - i, stg[i & 1], s0);\n                fetch(K - 4 - i);\n", "")'
$B b_nodrain 'rep("            drain(K - i, s1);\n", ""); rep("                drain(K - 1 - i, s0);\n", "")'
$B b_nopark 'rep("            park(K - 2 - i, stg[(i + 1) & 1], s1);\n            fetch(K - 3 - i);\n", ""); rep("                park(K - 3 - i, stg[i & 1], s0);\n                fetch(K - 4 - i);\n", "")'
$B b_nophase2 'rep("                bwd_u(Y, u); SB;\n", ""); rep("                bwd_u(X, u - 1); SB;\n", "")'
$B b_nophase1 'rep("                fwd_u(X, u); SB;\n", ""); rep("                if (u + 1 < TT / 4 - 1) { fwd_u(Y, u + 1); SB; }\n", ""); rep("            fwd_u(Y, TT / 4 - 1); SB;\n", "")'
