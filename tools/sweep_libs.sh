#!/bin/bash
# usage: tools/sweep_libs.sh "<variants>" "<configs>" <specs...>   (variants: '-' = default library)
V="$1"; C="$2"; shift 2
for c in $C; do for v in $V; do
  if [ "$v" = "-" ]; then unset SVSDF_LIB_VARIANT; else export SVSDF_LIB_VARIANT=$v; fi
  echo "== $c lib=$v"; python tools/sweep.py $c 1000000 "$@"
done; done
