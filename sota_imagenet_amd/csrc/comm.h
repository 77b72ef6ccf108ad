// comm.h — internal interface between the executor (resnet_exec.cpp) and the RCCL binding (comm.cpp).
#pragma once
#include "common.h"

struct mi355_comm;

namespace mi355 {
int comm_allreduce_bucket(mi355_comm* cm, float* grads, size_t begin, size_t end, hipStream_t s, hipStream_t side);
int comm_join(mi355_comm* cm, hipStream_t s);
}  // namespace mi355
