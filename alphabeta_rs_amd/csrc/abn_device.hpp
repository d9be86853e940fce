// All device code of libabneutral_hip.so (one translation unit per .hip file includes this).
#pragma once
#include "abn_common.hpp"
#include "abn_fit_kernel.hpp"
#include "abn_fit_refill.hpp"
#include "abn_fit_spec.hpp"
#include "abn_aux_kernels.hpp"
#include "abn_pairwise_mx.hpp"
