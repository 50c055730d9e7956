#!/bin/bash
# Builds build_ab/<name>.so from the csrc/ of a git revision (A/B partner of the in-tree library: SAH_HIP_LIBRARY=build_ab/<name>.so).
#   tools/experiments/r6/build_rev.sh <rev> <name>
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/../../.." && pwd)
SRC=$ROOT/build_ab/src_$NAME
rm -rf "$SRC"; mkdir -p "$SRC"
git -C "$ROOT" archive "$REV" androidrenderer_amd/csrc include | tar -x -C "$SRC"
SAH_HIP_CSRC=$SRC/androidrenderer_amd/csrc SAH_HIP_LIBRARY=$ROOT/build_ab/$NAME.so python -m androidrenderer_amd.build > "$ROOT/build_ab/$NAME.log" 2>&1
ls -la "$ROOT/build_ab/$NAME.so"
