#!/bin/bash
# Data-parallel label generation for the Cityscapes "train_extra" split (22 973 images read straight
# from the two zip archives): N_GPUS background processes, contiguous image ranges of size
# n_data / N_GPUS + 1, one GPU each (the reference's fan-out, utils/create_train_extra_labels.sh
# there, same parameters: felzenszwalb 300 / 0.8 / 20, 4 clusters, batches of 30).
#   usage: bash utils/create_train_extra_labels.sh N_GPUS [slic|felzenszwalb] [extra driver flags...]
# e.g. BASELINE config 5:  bash utils/create_train_extra_labels.sh 8 slic --n_slic_segments 400 \
#          --arch drn_d_22 --dtype bf16 --pool_mode mean --resize_shape 1024 2048 --no_figure
# One process per GPU under torchrun instead (rank 0 writes result.json after one RCCL all_gather):
#   python -m torch.distributed.run --nproc-per-node N_GPUS batch_spalign_kmeans.py <the same flags>
N_GPUS=${1:-1}
METHOD=${2:-felzenszwalb}
shift; shift
N_CLUSTERS=4
BATCHSIZE=30
OUT=results/estimated_train_extra_labels
n_data=22973
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
        --cityscapes_img_zip data/cityscapes_train_extra_imgs.0.zip \
        --cityscapes_label_zip data/cityscapes_train_extra_labels.0.zip --gpu 0 "$@" &
    gpu=$(( gpu + 1 ))
done
wait
