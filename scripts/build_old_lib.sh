#!/bin/bash
# Build libmcpc of an earlier commit into scripts/bin/libmcpc_<commit>.so (developer tool: A/B runs and scripts/nan_repro.py).
#   scripts/build_old_lib.sh 71cd49e b0072e4
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/bin
for c in "$@"; do
    tmp=$(mktemp -d)
    git archive "$c" montecarlopredictivecoding_amd/csrc include | tar -x -C "$tmp"
    make -s -C "$tmp/montecarlopredictivecoding_amd/csrc" OUT="$PWD/scripts/bin/libmcpc_$c.so"
    rm -rf "$tmp"
    ls -la "scripts/bin/libmcpc_$c.so"
done
