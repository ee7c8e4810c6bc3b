#ifndef QST_KERNELS_PUBLIC_H
#define QST_KERNELS_PUBLIC_H
/* Kernel-level entry points of libqst.so (building blocks of qst_encoder_forward/backward). */
#include "../quadruplet-sentence-transformer_amd/csrc/qst_kernels.h"
#endif
