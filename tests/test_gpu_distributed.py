"""GPU: two data-parallel ranks of the real modules on one GPU (gloo): see tests/dp_rehearsal.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_end_a_step_with_identical_parameters():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['OMP_NUM_THREADS'] = '4'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'dp_rehearsal.py')],
                       capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    if r.returncode != 0:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'dp_rehearsal.err'), 'w') as fh:
            fh.write(r.stdout + '\n---- stderr ----\n' + r.stderr)
        tb = [l for l in r.stderr.splitlines() if 'Error' in l or 'error' in l or l.startswith('  File')]
        raise AssertionError('dp_rehearsal failed (full output: gpurun_out/dp_rehearsal.err):\n' + '\n'.join(tb[-25:]))
    line = [l for l in r.stdout.splitlines() if l.startswith('DP_REHEARSAL ')][-1]
    res = json.loads(line[len('DP_REHEARSAL '):])
    print(json.dumps(res))
    assert res['eager_grad_err'] < 1e-6 and res['graph_grad_err'] < 1e-6, res
    assert res['eager_params_identical'] and res['graph_params_identical'], res
    assert res['eager_stats'] == {'early_buckets': 3, 'late_buckets': 0}, res     # tail | blocks4 | final, all from inside backward
    assert res['graph_segments'] == 4, res                     # tail | blocks4 | final | autograd hand-over
    # the discriminator alone: every parameter gradient is the mean of the ranks' local sums over the two passes, and the head's weight
    # gradient got there as two locally formed products of the gathered factors (one per pass), eager and replayed
    for mode in ('eager', 'graph'):
        assert res['d_%s_grad_err' % mode] < 1e-5, res
        assert res['d_%s_factored_products' % mode] == 2, res
        assert res['d_%s_stats' % mode]['late_buckets'] == 0, res
    # the joint D + G iteration: both networks end identical on both ranks, eager and replayed; the discriminator's two backward
    # passes of the D step each announced head / convs / final from inside the schedule, its G-step pass announced nothing
    for mode in ('eager', 'graph'):
        assert res['joint_%s_g_identical' % mode] and res['joint_%s_d_identical' % mode], res
        assert res['joint_%s_losses_finite' % mode], res
        assert res['joint_%s_d_stats' % mode]['late_buckets'] == 0 and res['joint_%s_d_stats' % mode]['early_buckets'] >= 2 * 2 * 3, res
    assert res['joint_graph_segments'] >= 8, res
