"""The parts of bench.py that do not need a GPU: argument contract (--gpus / --steps / --warmup with defaults that
finish in minutes) and the cpu_baseline leg (numpy oracle on a row sample, extrapolated)."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_argument_contract(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    a = bench.parse()
    assert a.gpus == 1 and a.steps == 20 and a.warmup == 3 and int(a.n_obs) == 1_000_000 and a.n_free == 1024
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '5', '--warmup', '2'])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2)


def test_cpu_baseline_leg_on_a_small_sample():
    bench = _bench()
    rng = np.random.default_rng(0)
    ns, D, n_pos = 256, 32, 8
    x = rng.normal(size=(ns, D)) / np.sqrt(D)
    y = rng.normal(size=ns)
    out = bench.cpu_baseline(x, y, n_total=4096, D=D, n_pos=n_pos, loss='gaussian', lik_info=2.0, prior_info=1.0,
                             theta=rng.normal(size=D) * 0.05)
    assert out['kind'] == 'port' and out['unit'] == 'hessian_builds/s' and out['cores'] >= 1
    assert out['value'] > 0 and out['strong_numpy_value'] > 0 and 'rows' in out['sample']
    # the port really is the D-pass structure: it cannot beat the closed form on the same sample
    assert out['value'] <= out['strong_numpy_value'] * 1.5


def test_stdout_carries_only_the_result_line(monkeypatch, capfd):
    """Libraries underneath the bench (RCCL's version banner) write to file descriptor 1; the wrapper must
    send all of that to stderr and put exactly the JSON line on stdout."""
    import json
    bench = _bench()

    def noisy_main():
        os.write(1, b'banner from a native library\n')           # what librccl does at communicator start-up
        return {'metric': 'm', 'value': 1.0}

    monkeypatch.setattr(bench, 'main', noisy_main)
    bench.run_with_clean_stdout()
    out, err = capfd.readouterr()
    assert out.count('\n') == 1 and json.loads(out) == {'metric': 'm', 'value': 1.0}
    assert 'banner from a native library' in err
    # ranks other than 0 return nothing and print nothing
    monkeypatch.setattr(bench, 'main', lambda: None)
    saved = os.dup(1)
    try:
        bench.run_with_clean_stdout()
        out, _ = capfd.readouterr()
        assert out == ''
    finally:
        os.dup2(saved, 1)
        os.close(saved)
