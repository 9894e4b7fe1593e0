#!/bin/bash
# usage: tools/build_ref_variant.sh <commit> <tag> [-DFLAG ...]
# Builds implicit-svsdf-planner_amd/libsvsdf_hip_<tag>.so (a -DSVSDF_FAST_BUILD variant: star / sdHorseshoe / sdHeart /
# Polygon) from the sources of <commit>, in a temporary git worktree -- the baseline for an A/B with tools/exp_variants.py
# or tools/poly_outline_ab.py.  (The default library is NOT a baseline: tests/conftest.py's `built` fixture and
# __graft_entry__.build() rebuild it from the working tree whenever a source is newer.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
COMMIT=${1:?commit}; TAG=${2:?tag}; shift 2
WT=$(mktemp -d /tmp/svsdf_ref_XXXXXX)
git -C "$ROOT" worktree add -f "$WT" "$COMMIT" > /dev/null
trap 'git -C "$ROOT" worktree remove --force "$WT" > /dev/null 2>&1; git -C "$ROOT" worktree prune' EXIT
(cd "$WT/implicit-svsdf-planner_amd" && python build.py --variant "$TAG" "$@" > /dev/null)
cp "$WT/implicit-svsdf-planner_amd/libsvsdf_hip_$TAG.so" "$ROOT/implicit-svsdf-planner_amd/"
echo "$ROOT/implicit-svsdf-planner_amd/libsvsdf_hip_$TAG.so  <- $(git -C "$ROOT" rev-parse --short "$COMMIT")"
