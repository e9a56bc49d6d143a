#!/bin/bash
# ablation set 2: the forward's scan wave ALONE (staging waves only keep the barriers) -- what does its own stream cost?
set -e
B=tools/scan_exp/build_variant.sh
NS='rep("            if (!STATE_ONLY) finish(k - 1, k0);\n            park(k + 1, k0);\n            fetch(k + 2);\n", ""); rep("                if (!STATE_ONLY) finish(k, k1);\n                park(k + 2, k1);\n                fetch(k + 3);\n", "");'
NOY='rep("                if (!STATE_ONLY) yp[(4 * u + s) * 8 * PS] = fmaf(h.y, U.bc[s].w, h.x * U.bc[s].z);\n", "");'
NOEXP='rep("                a[s] = f2{fast_exp2(x.x), fast_exp2(x.y)};\n                xb[s] = f2{U.bc[s].x, U.bc[s].y} * dtu;", "                a[s] = x;\n                xb[s] = f2{U.bc[s].x, U.bc[s].y} * dtu;");'
NOLD='rep("            load_u(ub, u + 1); SB;\n            step_u(ua, u); SB;\n            if (u + 2 < TT / 4) load_u(ua, u + 2);\n            SB;\n            step_u(ub, u + 1); SB;\n", "            asm volatile(\"\" : \"+v\"(ua.dd[0]), \"+v\"(ua.dd[1]), \"+v\"(ua.bc[0]), \"+v\"(ua.bc[1]), \"+v\"(ua.bc[2]), \"+v\"(ua.bc[3])); step_u(ua, u); SB;\n            asm volatile(\"\" : \"+v\"(ua.dd[0]), \"+v\"(ua.dd[1]), \"+v\"(ua.bc[0]), \"+v\"(ua.bc[1]), \"+v\"(ua.bc[2]), \"+v\"(ua.bc[3])); step_u(ua, u + 1); SB;\n");'
NOWR='rep("                if (!STATE_ONLY) yp[(4 * u + s) * 8 * PS] = fmaf(h.y, U.bc[s].w, h.x * U.bc[s].z);\n", "                ysum += fmaf(h.y, U.bc[s].w, h.x * U.bc[s].z);\n"); rep("        U4 ua, ub;\n        S2_STAMP(1)\n        load_u(ua, 0);", "        U4 ua, ub;\n        S2_STAMP(1)\n        load_u(ua, 0);\n        if (k == nt - 1) yp[0] = ysum;"); rep("    lds_barrier();\n    S2_STAMP_DECL\n    for (int k = 0; k < nt; ++k) {\n        S2_STAMP(0)\n        if (do_ck)", "    lds_barrier();\n    float ysum = 0.f;\n    S2_STAMP_DECL\n    for (int k = 0; k < nt; ++k) {\n        S2_STAMP(0)\n        if (do_ck)");'
$B s0 "$NS"
$B s1_noy "$NS $NOY"
$B s2_noy_nold "$NS $NOY $NOLD"
$B s3_noy_noexp "$NS $NOY $NOEXP"
$B s4_nowr "$NS $NOWR"
$B s5_nold "$NS $NOLD"
$B s6_noy_noexp_nold "$NS $NOY $NOEXP $NOLD"
