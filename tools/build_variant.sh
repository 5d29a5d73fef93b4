#!/bin/bash
# Builds the library of a git revision into defl_amd/variants/libdefl_<name>.so (for A/B timing with tools/ab.sh):
#   tools/build_variant.sh <revision> <name> [EXTRA flags]
set -e
cd "$(dirname "$0")/.."
rev=$1; name=$2; shift 2
tmp=$(mktemp -d /tmp/variant.XXXXXX)
git archive "$rev" defl_amd/csrc include | tar -x -C "$tmp"
mkdir -p defl_amd/variants
make -s -C "$tmp/defl_amd/csrc" -j4 OUT="$PWD/defl_amd/variants/libdefl_$name.so" EXTRA="$*" 2>&1 | grep -E "error" || true
rm -rf "$tmp"
ls -la "defl_amd/variants/libdefl_$name.so"
