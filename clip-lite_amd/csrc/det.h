// Deterministic-reduction mode (clite_set_deterministic, include/clite.h): a process-wide host flag read by every launcher.
// When it is on, no address receives float-atomic contributions from two workgroups (or two waves) of one launch, so every sum is
// formed in one fixed order and two runs of the same step are bit-identical:
//   * split-K GEMMs run unsplit (one workgroup owns an output tile over the whole K range); the workspace split-K is off
//   * column statistics fused into a GEMM epilogue are produced by colstats_det_kernel from the stored tensor instead, one row slab per
//     statistics replica
//   * row-slab reductions (BatchNorm backward, bias gradients, LayerNorm parameter gradients, gradient norm) use at most one
//     workgroup per accumulator replica; the loss-head kernels run as a single wave; the embedding backward as a single workgroup
// It is a debugging / testing aid (checkpoint-resume and graph-replay equivalence are exact under it) and costs several x in step time.
#ifndef CLITE_DET_H
#define CLITE_DET_H
namespace clite {
bool deterministic();
}
#endif
