"""Stale evidence must not pass silently: every entry of profiles/{bgk_traffic,gp_counters,side_counters}.json has to be
stamped with the hash of the kernel sources THIS tree is built from (bench.kernel_source_hash: the named header plus
everything it includes, e.g. sincos_table.inc and — for the GP / LV kernels — bgk_kernels.h).  bench.py refuses counters
with another stamp (roofline.traffic = null); this test makes that state a red test instead of a quiet null.
Regenerate after the last kernel edit:  gpurun -- 'bash tools/prof/restamp_all.sh'."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _entries():
    for path in sorted(bench.STAMPED):
        with open(os.path.join(ROOT, "profiles", path)) as f:
            d = json.load(f)
        assert d, path
        for key, e in d.items():
            yield path, key, e


@pytest.mark.parametrize("path,key,e", list(_entries()), ids=lambda v: v if isinstance(v, str) else "")
def test_entry_is_stamped_with_this_trees_kernel_sources(path, key, e):
    srcs = bench.stamped_sources(path, key)
    assert e.get("kernel_sha") == bench.kernel_source_hash(srcs), (
        f"profiles/{path}[{key}] was recorded with other kernel sources than {bench.kernel_source_files(srcs)}: "
        "re-run tools/prof/restamp_all.sh on the GPU box and commit the result")


def test_source_closure_follows_includes():
    assert "sincos_table.inc" in bench.kernel_source_files(("bgk_kernels.h",))
    f = bench.kernel_source_files(("gp_kernels.h",))
    assert "bgk_kernels.h" in f and "sincos_table.inc" in f
    f = bench.kernel_source_files(("lv_kernels.h", "devmap_lv_kernels.h"))
    assert "bgk_kernels.h" in f and "devmap_kernels.h" in f


def test_bench_legs_ask_for_the_sources_the_stamps_table_names():
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the legs pass their source tuples explicitly; they must agree with bench.STAMPED
    assert 'sources=("gp_kernels.h",)' in src and bench.stamped_sources("gp_counters.json", "gp_rays50000_d3") == ("gp_kernels.h",)
    assert 'sources=("lv_kernels.h", "devmap_lv_kernels.h")' in src
    assert 'sources=("bgkl_kernels.h",)' in src
