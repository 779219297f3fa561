"""CPU oracle for the point-cloud hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / the timed CPU baseline.  ``pointcloudlib_amd`` never imports it.

PARITY UNPINNED (see ``pcl_oracle.c`` header and DESIGN.md): the reference holds no golden vectors and
cannot be imported or compiled here; the oracle restates the reference's CUDA text and is pinned only
by hand-derived known answers and by a second, independent NumPy restatement.
"""
from .oracle import (  # noqa: F401
    lib_path, build, optimal_block, fps, ball_query, group, group_bwd, group_all, knn, three_nn,
    three_interp, num_threads, density, set_contract, get_contract, contract, knn_point_matmul,
)
