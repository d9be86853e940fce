// The device code of the fit path (abn_api.hip includes this; the pairwise scan of abn_pairwise.hip has its own header,
// abn_pairwise_mx.hpp).
#pragma once
#include "abn_common.hpp"
#include "abn_fit_kernel.hpp"
#include "abn_fit_refill.hpp"
#include "abn_fit_spec.hpp"
#include "abn_aux_kernels.hpp"
