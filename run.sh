#!/bin/bash
# The reference's run.sh (SchlomoFeng/GDN run.sh:4-58) for gdn_amd: same hyper-parameters, same argument
# order `bash run.sh <gpu_n> <dataset>`; the reference's `cpu` first argument has no counterpart (no CPU path).
gpu_n=$1
DATASET=$2

seed=5
BATCH_SIZE=32
SLIDE_WIN=5
dim=64
out_layer_num=1
SLIDE_STRIDE=1
topk=5
out_layer_inter_dim=128
val_ratio=0.2
decay=0

path_pattern="${DATASET}"
COMMENT="${DATASET}"

EPOCH=30
report='best'

if [[ "$gpu_n" == "cpu" ]]; then
    echo "gdn_amd runs on an MI355X only; use the reference for a CPU run" >&2
    exit 2
fi
CUDA_VISIBLE_DEVICES=$gpu_n HIP_VISIBLE_DEVICES=$gpu_n python -m gdn_amd.main \
    -dataset $DATASET \
    -save_path_pattern $path_pattern \
    -slide_stride $SLIDE_STRIDE \
    -slide_win $SLIDE_WIN \
    -batch $BATCH_SIZE \
    -epoch $EPOCH \
    -comment $COMMENT \
    -random_seed $seed \
    -decay $decay \
    -dim $dim \
    -out_layer_num $out_layer_num \
    -out_layer_inter_dim $out_layer_inter_dim \
    -decay $decay \
    -val_ratio $val_ratio \
    -report $report \
    -topk $topk
