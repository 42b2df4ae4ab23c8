// dataset.hpp -- the feature matrix resident in HBM, shared by vae.hip (training / encoding) and prep.hip (the
// device side of make_dataloader).
#pragma once

#include "common.hpp"

constexpr int kDatasetColPad = 32;   // columns of X are padded to a multiple of this (the GEMM K-tiles)

// The feature matrix [n][D_p] (zero padded; columns: S depths | 103 TNF | 1 total abundance) + weights [n], resident
// in HBM.  Owned by a VAE handle (vh_vae_set_dataset) or shared between handles (vh_dataset_create / vh_prep_finish
// + vh_vae_use_dataset): the dataset of one `vamb bin default` run is uploaded once however many models are trained on it.
// Semi-supervised models (semisupervised_encode.py:111-175): one int32 class per row (vh_dataset_set_labels); the one-hot
// columns are produced by the batch gather, never stored.  A labels-only dataset (vh_dataset_create_labels) has no
// feature matrix: S = D_p = 0, unit weights.
struct vh_dataset {
    vh::DevBuf<float> X, w;
    vh::DevBuf<int32_t> labels;
    int64_t n = 0;
    int S = 0, D_p = 0;
    int NL = 0;   // width of the one-hot label block (0: no labels)
    int32_t max_label = -1;   // largest label present (a hierarchical loss needs every label to be a node of its tree)
};
