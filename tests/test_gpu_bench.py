"""-m gpu: bench.py as the driver runs it.  `python bench.py --gpus N` must start its N ranks itself (it used to
exit unless wrapped in torch.distributed.run), print exactly one JSON line on stdout, and say how the ranks talked.  On a
one-GPU box the ranks share cuda:0 over gloo -- the same code path as the RCCL run minus the wire.  `--as-rank R/P` is the
one-GPU stand-in for rank R of a P-GPU job (collectives stubbed) that tools/scale_model.py feeds on."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['--scale', '0.01', '--steps', '3', '--warmup', '2', '--no-pmc', '--no-cpu-baseline', '--no-epoch', '--no-kernel-timing',
         '--batch-size', '64']


def _bench(*extra, env=None, timeout=900):
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *SMALL, *extra], cwd=ROOT, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f'stdout must carry exactly one line, got {len(lines)}: {p.stdout[:2000]}'
    return json.loads(lines[0])


def test_bench_one_gpu_line_has_the_contract_keys():
    out = _bench('--gpus', '1')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in out, k
    assert out['n_gpus'] == 1 and out['steps'] == 3 and out['value'] > 0
    ov = out['config']['sampler_overlap']
    assert ov is not None and 'overlap_ratio' in ov          # the line says whether the side sampler really ran beside the step


@pytest.mark.parametrize('extra', [[], ['--scaling', 'strong'], ['--parallelism', 'shard']], ids=['weak-seed', 'strong-seed', 'shard'])
def test_bench_starts_its_own_ranks(extra):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: two ranks, one line, rank 0's."""
    out = _bench('--gpus', '2', *extra)
    assert out['n_gpus'] == 2
    comm = out['config']['communication']
    assert comm['world_size'] == 2 and comm['backend'] in ('gloo', 'nccl')
    assert comm['collectives_per_step_and_rank'], 'a two-rank step moves gradients: the line must list its collectives'
    if comm['backend'] == 'gloo':
        assert 'SHARE' in comm['note'] and out['config']['rccl_world_size'] == 0
    assert out['value'] > 0 and out['ms_per_step'] > 0
    assert out['scaling'] == ('weak' if not extra else 'strong')
    # a first real N > 1 run explains itself -- measured ms of every collective next to the alpha-beta model's,
    # every rank's own event-timed step, the rank count as the backend's own collective sees it, the scaling statement
    meas = comm['collectives_measured']
    assert meas and all(v['ms_per_call'] > 0 and v['calls_per_step'] > 0 and v['model_ms_per_call_direct'] > 0 and
                        v['model_ms_per_call_ring'] > 0 for v in meas.values()), meas
    assert any(k.startswith('all_reduce(') for k in meas)
    pr = comm['per_rank_ms']
    assert len(pr['by_rank']) == 2 and 0 < pr['min'] <= pr['max'] <= 1.5 * out['ms_per_step'] + 1.0
    assert comm['ranks_seen_by_the_backend'] == 2
    assert 'north_star' in out['scaling_note'] and 'weak' in out['scaling_note']


@pytest.mark.parametrize('extra', [[], ['--scaling', 'strong'], ['--parallelism', 'shard']], ids=['weak-seed', 'strong-seed', 'shard'])
def test_bench_as_rank_emulates_one_rank_of_eight(extra):
    out = _bench('--as-rank', '3/8', *extra)
    assert out['n_gpus'] == 1 and out['config']['emulated_rank'] == {'rank': 3, 'world': 8}
    comm = out['config']['communication']
    assert comm['backend'] == 'fake' and comm['world_size'] == 8
    assert comm['collectives_per_step_and_rank'], 'the emulated rank must list the collectives (and bytes) it would issue'
    assert out['ms_per_step'] > 0
