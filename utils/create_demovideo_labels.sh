#!/bin/bash
# Label-free road masks for the Cityscapes demoVideo frames (2 899 images listed in
# data/demoVideo_fns.txt): N_GPUS background processes of utils/apply_spalign_kmeans.py, contiguous
# ranges of size n_data / N_GPUS + 1, one GPU each (the reference's utils/create_demovideo_labels.sh,
# same parameters; its first --out_dir is overridden by the second, results/estimated_demoVideo_labels).
#   usage: bash utils/create_demovideo_labels.sh N_GPUS [slic|felzenszwalb] [extra driver flags...]
N_GPUS=${1:-1}
METHOD=${2:-felzenszwalb}
shift; shift
N_CLUSTERS=4
BATCHSIZE=30
OUT=results/estimated_demoVideo_labels
n_data=2899
step=$(( n_data / N_GPUS + 1 ))
gpu=0
for (( s=0; s<n_data; s+=step )); do
    e=$(( s + step < n_data ? s + step : n_data ))
    HIP_VISIBLE_DEVICES=$gpu PYTHONWARNINGS=ignore python utils/apply_spalign_kmeans.py \
        --superpixel_method $METHOD --n_slic_segments 100 --n_clusters $N_CLUSTERS \
        --y_rel_pos 0.75 --x_rel_pos 0.5 --y_rel_sigma 0.1 --x_rel_sigma 0.1 \
        --n_anchors 10 --n_neighbors 4 --batchsize $BATCHSIZE \
        --felzenszwalb_scale 300 --felzenszwalb_sigma 0.8 --felzenszwalb_min_size 20 \
        --use_feature_maps 7 --start_index $s --end_index $e \
        --img_list_fn data/demoVideo_fns.txt --out_dir $OUT --gpu 0 "$@" &
    gpu=$(( gpu + 1 ))
done
wait
