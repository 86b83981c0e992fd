// Job tables of the batched weight preparation (include/pfst_hip.h pfst_weight_job_t): which job a workgroup belongs to, and the host-side
// check of a table before a launch that will trust it.
#pragma once
#include "common.h"
#include "../../include/pfst_hip.h"

// the last job whose first_block <= b (first_block is a non-decreasing prefix; jobs[0].first_block == 0)
__device__ __forceinline__ int weight_job_of_block(const pfst_weight_job_t* __restrict__ jobs, int njobs, int b) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// workgroups of one job: prep -- 4096 floats (m == 0) or 256 filters (Winograd) per workgroup; pack -- 256 16-byte chunks per
// workgroup, whole workgroups per filter set (the scale is wave-uniform)
static inline i64 weight_job_pack_chunks(const pfst_weight_job_t& j) {       // chunks of ONE set
  const i64 nf = j.dst_f ? (i64)(j.T * j.Cin / 16) * 2 * j.Cout : 0, nd = j.dst_d ? (i64)(j.T * j.Cout / 16) * 2 * j.Cin : 0;
  return nf + nd;
}
static inline int weight_job_blocks(const pfst_weight_job_t& j, int pack) {
  if (pack) return j.sets * cdiv(weight_job_pack_chunks(j), 256);
  return j.m == 0 ? cdiv((i64)j.Cout * j.Cin * j.T, 4096) : cdiv((i64)j.Cout * j.Cin, 256);
}
