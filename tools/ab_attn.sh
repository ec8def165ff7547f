#!/bin/bash
# On the GPU box: time the config-3 attention launch (B=1) with each variant library named on the command line
# ("cur" = the in-tree library).   bash tools/ab_attn.sh cur old L0G1 ...
for v in "$@"; do
  if [ "$v" = cur ]; then unset EA_LIB_PATH; else export EA_LIB_PATH=$PWD/easyanimate_amd/lib/variants/libea_$v.so; fi
  r=$(python tools/ab_attn_lib.py 2>&1 | grep 'attention v3' | sed 's/.*"TFLOPs": \([0-9.]*\).*/\1/' | tr '\n' ' ')
  echo "variant=$v $r"
done
