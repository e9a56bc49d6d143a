#!/bin/bash
set -e
B=tools/scan_exp/build_variant.sh
NS='rep("            if (!STATE_ONLY) finish(k - 1, k0);\n            park(k + 2, 0, k0);\n            fetch(k + 3);\n", ""); rep("                if (!STATE_ONLY) finish(k, k1);\n                park(k + 3, 1, k1);\n                fetch(k + 4);\n", ""); rep("                if (!STATE_ONLY) finish(k + 1, k2);\n                park(k + 4, 2, k2);\n                fetch(k + 5);\n", "");'
$B base 'pass'
$B s0 "$NS"
