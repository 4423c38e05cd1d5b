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
    N, D, n_pos = 30000, 32, 8
    x = rng.normal(size=(N, D)) / np.sqrt(D)
    y = rng.normal(size=N)
    out = bench.cpu_baseline(lambda a, b: (x[a:b], y[a:b]), n_total=N, D=D, n_pos=n_pos, loss='gaussian', lik_info=2.0,
                             prior_info=1.0, theta=rng.normal(size=D) * 0.05, budget_s=2.0)
    assert out['kind'] == 'port' and out['unit'] == 'hessian_builds/s' and out['cores'] >= 1
    assert out['value'] > 0 and out['strong_numpy_value'] > 0 and 'rows' in out['sample']
    raw = out['raw_timings']
    # the strong path is a measurement over ALL rows, not an extrapolation; the port path is timed at two sizes
    assert raw['strong']['rows_timed'] == N and raw['strong']['gflops'] > 0
    assert raw['port_n10000']['rows'] == 10000 and raw['port_n30000']['rows'] == 30000
    assert raw['port_fit']['per_column_per_row_s'] >= 0 and raw['port_fit']['per_column_fixed_s'] >= 0
    # the port really is the D-pass structure: it cannot beat the closed form
    assert out['value'] <= out['strong_numpy_value'] * 1.5
    # the checker leg of the `parity` record: the matrix the strong leg forms over all rows IS the oracle's Hessian
    from oracle import packing as opk, models as om
    theta = np.random.default_rng(0)
    rng2 = np.random.default_rng(0); rng2.normal(size=(N, D)); rng2.normal(size=N); theta = rng2.normal(size=D) * 0.05
    lay = opk.Layout([opk.box_block(D - n_pos), opk.box_block(n_pos, lb=0.0)])
    model = om.DeclaredModel(lay, loss=om.GAUSSIAN, x=x, y=y, lik_info=2.0, quad_A=np.full(D, 1.0))
    full = out['_oracle_full']
    Hm = model.hessian(theta)
    assert np.max(np.abs(full['hessian'] - Hm)) < 1e-12 * np.max(np.abs(Hm))
    assert np.max(np.abs(full['grad'] - model.grad(theta))) < 1e-12 * np.max(np.abs(model.grad(theta)))


def test_gpus_flag_starts_the_ranks_itself(monkeypatch):
    """`python bench.py --gpus N` (no rank environment) must start N ranks as a child torchrun; inside a rank
    (RANK / WORLD_SIZE set, as torchrun and the driver's own launcher set them) it must not."""
    bench = _bench()
    a = bench.parse(['--gpus', '8', '--steps', '5', '--warmup', '2'])
    assert bench.needs_launch(a, {}) and not bench.needs_launch(a, {'RANK': '3', 'WORLD_SIZE': '8'})
    assert not bench.needs_launch(bench.parse([]), {})
    cmd = bench.launch_command(8, ['--gpus', '8', '--steps', '5', '--warmup', '2'], 29555)
    assert cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nproc-per-node=8' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29555'
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '8', '--steps', '5', '--warmup', '2']


def test_launcher_relays_one_line_and_the_exit_code(monkeypatch, capfd):
    """The parent relays exactly rank 0's JSON line and fails loudly when the ranks fail."""
    import subprocess
    bench = _bench()
    a = bench.parse(['--gpus', '2'])

    class Done(object):
        def __init__(self, rc, out):
            self.returncode, self.stdout = rc, out
    monkeypatch.setattr(subprocess, 'run', lambda cmd, **kw: Done(0, b'banner\n{"metric": "m", "n_gpus": 2}\n'))
    assert bench.launch_ranks(a, ['--gpus', '2']) == 0
    out, _ = capfd.readouterr()
    assert out == '{"metric": "m", "n_gpus": 2}\n'
    monkeypatch.setattr(subprocess, 'run', lambda cmd, **kw: Done(1, b''))
    assert bench.launch_ranks(a, ['--gpus', '2']) != 0
    out, err = capfd.readouterr()
    assert out == '' and 'failed' in err


def test_a_rank_count_mismatch_is_an_error(monkeypatch):
    """--gpus N inside a world of another size is refused (it used to print a warning and measure one GPU)."""
    import pytest
    bench = _bench()
    monkeypatch.setenv('WORLD_SIZE', '2'); monkeypatch.setenv('RANK', '0')
    with pytest.raises(SystemExit):
        bench.main(bench.parse(['--gpus', '4']))


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
