#!/bin/bash
# Data-parallel label generation for the Cityscapes "val" list: N_GPUS background processes,
# contiguous image ranges of size n_data / N_GPUS + 1, one GPU each (the reference's fan-out,
# utils/create_val_labels.sh there).  Defaults reproduce the reference launcher's parameters
# (felzenszwalb superpixels, scale 300, as in the reference; pass "slic" as 2nd argument for SLIC).
#   usage: bash utils/create_val_labels.sh N_GPUS [slic|felzenszwalb]
N_GPUS=${1:-1}
METHOD=${2:-felzenszwalb}
N_CLUSTERS=4
BATCHSIZE=30
OUT=results/estimated_val_labels
n_data=500
step=$(( n_data / N_GPUS + 1 ))
gpu=0
for (( s=0; s<n_data; s+=step )); do
    e=$(( s + step < n_data ? s + step : n_data ))
    HIP_VISIBLE_DEVICES=$gpu PYTHONWARNINGS=ignore python batch_spalign_kmeans.py \
        --superpixel_method $METHOD --n_slic_segments 100 --n_clusters $N_CLUSTERS \
        --y_rel_pos 0.75 --x_rel_pos 0.5 --y_rel_sigma 0.1 --x_rel_sigma 0.1 \
        --n_anchors 10 --n_neighbors 4 --batchsize $BATCHSIZE \
        --felzenszwalb_scale 300 --felzenszwalb_sigma 0.8 --felzenszwalb_min_size 20 \
        --use_feature_maps 7 --out_dir $OUT --start_index $s --end_index $e \
        --img_file_list data/val_images.txt --label_file_list data/val_labels.txt --gpu 0 &
    gpu=$(( gpu + 1 ))
done
wait
