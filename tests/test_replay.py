"""The single-GPU replay harness (tools/replay.py, csrc/replay/replay_comm.hip -> libcap_replay.so): a measurement aid ABOVE the C ABI - it
only calls cap_comm_create_callbacks - that lets one GPU run the real schedule of one rank of the 1 x P plan with the peers' contributions
copied out of a finished factor.  Not gpu: it builds, stays out of libcapital_amd.so and exports its entry points.
gpu: a small replay reproduces the single-GPU factor on the replayed ranks' columns."""
import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_replay_library_builds_beside_the_product_and_exports_its_entry_points():
    from capital_amd import build
    lib = build.build_replay(verbose=False)                  # hipcc cross-compiles without a GPU
    assert os.path.exists(lib) and os.path.basename(lib) == "libcap_replay.so"
    L = ctypes.CDLL(build.LIB, mode=ctypes.RTLD_GLOBAL)      # the product first: the harness links against it
    R = ctypes.CDLL(lib)
    for name in ("cap_replay_create", "cap_replay_stats", "cap_replay_set_strip", "cap_replay_set_channels", "cap_replay_destroy", "cap_replay2d_create",
                 "cap_replay2d_stats", "cap_replay2d_set_strip", "cap_replay2d_destroy"):
        assert hasattr(R, name), name
        assert not hasattr(L, name), "the replay harness must stay out of the product library: " + name
    # the product-side aid it relies on is an ordinary plan option
    assert b"remote_chain_us" in open(build.LIB, "rb").read()


@pytest.mark.gpu
def test_small_replay_reproduces_the_single_gpu_factor():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import replay
    res = replay.run(n=4096, P=4, ranks=[0, 3], nb=512, steps=1, warmup=1, link_GBps=100.0, lat_us=5.0)
    assert len(res["ranks"]) == 2 and res["projected_ms_max_over_ranks"] > 0
    for r in res["ranks"]:
        assert r["R_max_abs_diff_vs_single_gpu"] <= 1e-11 * r["R_max_abs"], r       # another blocking of the same sums: to rounding
        assert r["GB_from_peers_per_step"] > 0 and r["link_model_ms_per_step"] > 0
        assert set(r["busy_ms"]) == {"chains", "row_solves", "head_updates", "msg_broadcasts", "strip_exchanges", "bulk_updates"}


@pytest.mark.gpu
def test_small_2d_replay_reproduces_the_single_gpu_factor():
    """the Pr x Pc plan on replay communicators whose broadcasts are generated from the plan's own loop: a call out of sequence (count, root)
    fails the factor call, a wrong payload gives a wrong piece of R"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import replay
    res = replay.run2d(n=4096, Pr=2, Pc=4, ranks=[0, 3, 5, 6], nb=512, steps=1, warmup=1)
    assert len(res["ranks"]) == 4
    for r in res["ranks"]:
        assert r["R_max_abs_diff_vs_single_gpu"] <= 1e-11 * 128.0, r
        assert r["GB_from_peers_per_step"] > 0
