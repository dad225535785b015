"""CPU: the bookkeeping around bench.py's static counters - the committed PMC summary names the kernel sources it was measured on, bench.py marks it
stale when they differ."""
import importlib.util
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_source_hash_follows_the_kernel_sources(tmp_path):
    pmc = _load('pmc_summary', os.path.join(ROOT, 'tools', 'pmc_summary.py'))
    root = tmp_path / 'r'
    shutil.copytree(os.path.join(ROOT, 'fql_amd', 'csrc'), root / 'fql_amd' / 'csrc')
    a = pmc.source_sha16(str(root))
    assert a == pmc.source_sha16(ROOT) and len(a) == 16
    with open(root / 'fql_amd' / 'csrc' / 'fql_aux.h', 'a') as f:
        f.write('\n// edited\n')
    assert pmc.source_sha16(str(root)) != a


def test_committed_pmc_summary_belongs_to_this_tree_and_bench_reads_it():
    bench = _load('bench_mod', os.path.join(ROOT, 'bench.py'))
    pmc = _load('pmc_summary', os.path.join(ROOT, 'tools', 'pmc_summary.py'))
    path = os.path.join(ROOT, 'profiles', bench.PMC_FILE)
    assert os.path.exists(path), 'run tools/profile.sh and copy its summary into profiles/'
    pj = json.load(open(path))
    got = bench.pmc_counters(bench.PMC_FILE, 'fql_side_kernel')
    assert got['traffic'] == pj['kernels']['fql_side_kernel']['hbm_bytes_per_launch'] > 0
    assert got['traffic_source_sha16'] == pj['source_sha16']
    assert got['traffic_stale'] == (pj['source_sha16'] != pmc.source_sha16(ROOT))
    if got['traffic_stale']:   # not an error (bench.py says so in its line), but worth seeing in the test log
        import warnings
        warnings.warn('fql_amd/csrc changed after profiles/%s was measured: re-run tools/profile.sh' % bench.PMC_FILE)
    assert bench.pmc_counters('no_such_file.json', 'fql_side_kernel') == {}
