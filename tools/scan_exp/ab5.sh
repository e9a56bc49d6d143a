#!/bin/bash
# ablation set 5 (final round-6 kernel): which of the late forward tweaks carries the 5 %: partial-row stride (36: conflict-free single stores; 40: paired ds_write2st64, 2-way conflicts), prefetch depth
set -e
B=tools/scan_exp/build_variant.sh
$B base 'pass'
$B ps36 'rep("constexpr int PS = 40; ", "constexpr int PS = 36; ")'
$B ld4 'rep("    constexpr int LD = 6;\n", "    constexpr int LD = 4;\n")'
$B ld8 'rep("    constexpr int LD = 6;\n", "    constexpr int LD = 8;\n")'
