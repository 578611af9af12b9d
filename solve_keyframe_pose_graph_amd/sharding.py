"""Edge sharding for multi-GPU solves: residual blocks are independent units, so each rank owns a
contiguous slice of every edge class; keyframe poses, CG vectors and the block-Jacobi preconditioner are replicated.
Regularisers live on rank 0.  Inside libpgo the only data-path collective is one RCCL all-reduce (fp64 sum) of the CG
matvec output per iteration, plus one of the diagonal blocks + gradient per linearisation."""
import numpy as np


def edge_slice(rank, world):
    """Returns the selector `sel(kind, n) -> index array` used by capi.problem_from_graph."""
    def sel(kind, n):
        if kind == "reg":
            return np.arange(n) if rank == 0 else np.arange(0)
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        return np.arange(lo, hi)
    return sel
