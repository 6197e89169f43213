"""-m gpu: the NON-default routings of the step, in the driver's own suite (housekeeping): every launch merge of
rounds 3-4 switched off (the unfused step the fused forms fall back to) and KGW_STRICT=1 (a route to the library GEMM is an
error).  The switches are read once per process, hence one subprocess per variant; each runs the committed golden-vector checks
and the captured-step-equals-eager test -- the oracle-pinned core -- under that environment."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MERGES_OFF = {'KGW_FUSED_ADAM': '0', 'KGW_MERGED_TRANSFORM_BWD': '0', 'KGW_DEFER_PRODUCTS': '0', 'KGW_DUV_PIECES': '0',
              'KGW_RELVEC_ALL': '0', 'KGW_MULTI_TRANSFORM': '0', 'KGW_ADAM_PACKS': '0', 'KGW_G3_RIDERS': '0', 'KGW_PARAM_TAIL': '0', 'KGW_DEFER_REDUCE': '0', 'KGW_DEFER_READOUT_FOLD': '0', 'KGW_PACK_FUSED': '0'}
VARIANTS = {'all_merges_off': MERGES_OFF, 'strict': {'KGW_STRICT': '1'}, 'unfolded_fc': {'KGW_FOLD_FC': '0'},
            'general_rows_only': {'KGW_SHORT_ROWS': '0', 'KGW_DUV_RIDERS': '0'},
            # (ONE switch off: the per-layer relation-vector node beside every other merge -- deferred fold backward, riding second
            #  launches -- still on)
            'relvec_per_layer': {'KGW_RELVEC_ALL': '0'}}


@pytest.mark.parametrize('variant', sorted(VARIANTS))
def test_fallback_routings_hold_the_golden_vectors(variant):
    env = dict(os.environ, **VARIANTS[variant])
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', os.path.join(ROOT, 'tests', 'test_gpu_golden.py'),
                        os.path.join(ROOT, 'tests', 'test_gpu_graph.py'), '-k', 'not measure_overlap'],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-3000:]
    assert ' passed' in p.stdout
