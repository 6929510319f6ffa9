'''svamd_cross_correlate_1d_static (the entry point that allocates nothing: its
workspace is an array inside the library) against svamd_cross_correlate_1d with a
caller-provided workspace, through the C ABI.  Two calls in a row, the second
with a shorter input: the shared array keeps the first call's contents, which
must not reach the second call's result.  Bar: bit-identical (float64 outputs of
the same kernels).  Needs an MI355X.
'''
import numpy as np
import pytest

import helpers  # noqa: F401

pytestmark = pytest.mark.gpu


def _both(nv, torch, v1, v2, normalised):
    n = len(v1)
    d1, d2 = nv.to_device(v1), nv.to_device(v2)
    want = nv.empty(2 * n - 1, torch.float64)
    nbytes = nv.load_library().svamd_cross_correlate_1d_workspace_bytes(n)
    ws = nv.workspace(nbytes)
    nv.call('svamd_cross_correlate_1d', nv.ptr(d1), nv.ptr(d2), n, normalised, nv.ptr(want), nv.ptr(ws),
            ws.numel(), nv.stream_ptr())
    got = nv.empty(2 * n - 1, torch.float64)
    got.fill_(float('nan'))
    # Device-level free memory (hipMemGetInfo) around the call: an allocation inside the
    # library would show here, which PyTorch's own allocator statistics cannot see.  The
    # first call on a device looks up the built-in array's address; `warm` keeps that
    # one-time step outside the window.
    _warm(nv, torch)
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()
    nv.call('svamd_cross_correlate_1d_static', nv.ptr(d1), nv.ptr(d2), n, normalised, nv.ptr(got),
            nv.stream_ptr())
    torch.cuda.synchronize()
    assert (torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()) == before
    return got.cpu().numpy(), want.cpu().numpy()


def _warm(nv, torch):
    v = nv.to_device(np.arange(8.0))
    out = nv.empty(15, torch.float64)
    nv.call('svamd_cross_correlate_1d_static', nv.ptr(v), nv.ptr(v), 8, 0, nv.ptr(out), nv.stream_ptr())
    torch.cuda.synchronize()


def test_static_workspace_1d_two_calls_in_a_row():
    import _native as nv
    torch = nv.torch_cuda()
    rng = np.random.default_rng(77)
    # first call: long enough for several template chunks and partial-sum planes
    n1 = 5000
    a = rng.standard_normal(n1) + 3.0
    b = np.roll(a, 137) + 0.1 * rng.standard_normal(n1)
    got, want = _both(nv, torch, a, b, 1)
    assert np.array_equal(got, want)
    assert int(np.argmax(got)) == int(np.argmax(want))
    # second call: shorter input, other flag; the array still holds the first call's data
    n2 = 700
    c = rng.random(n2)
    d = rng.random(n2)
    got, want = _both(nv, torch, c, d, 0)
    assert np.array_equal(got, want)
    # and against the reference's formula evaluated directly (correlation.py:64-73)
    padded = np.concatenate([np.zeros(n2 - 1), d, np.zeros(n2 - 1)])
    direct = np.array([np.sum((c - c.mean()) * (padded[i:i + n2] - d.mean())) / n2 for i in range(2 * n2 - 1)])
    assert np.max(np.abs(got - direct)) <= 1e-5 * np.max(np.abs(direct))


def test_static_workspace_1d_rejects_oversize():
    import _native as nv
    torch = nv.torch_cuda()
    n = 65536 + 1
    v = nv.to_device(np.zeros(n))
    out = nv.empty(2 * n - 1, torch.float64)
    with pytest.raises(nv.NativeError) as e:
        nv.call('svamd_cross_correlate_1d_static', nv.ptr(v), nv.ptr(v), n, 0, nv.ptr(out), nv.stream_ptr())
    assert e.value.code == nv.ERR_LIMIT
    assert 'limit' in str(e.value)
    # the largest accepted length runs
    n = 65536
    rng = np.random.default_rng(3)
    a, b = rng.random(n), rng.random(n)
    d1, d2 = nv.to_device(a), nv.to_device(b)
    got = nv.empty(2 * n - 1, torch.float64)
    nv.call('svamd_cross_correlate_1d_static', nv.ptr(d1), nv.ptr(d2), n, 0, nv.ptr(got), nv.stream_ptr())
    want = nv.empty(2 * n - 1, torch.float64)
    ws = nv.workspace(nv.load_library().svamd_cross_correlate_1d_workspace_bytes(n))
    nv.call('svamd_cross_correlate_1d', nv.ptr(d1), nv.ptr(d2), n, 0, nv.ptr(want), nv.ptr(ws), ws.numel(),
            nv.stream_ptr())
    assert torch.equal(got, want)


def test_static_workspace_1d_under_graph_capture():
    '''The use the entry point exists for: captured into a HIP graph (no allocation is
    possible there) and replayed, twice with different inputs copied into the captured
    buffers; replays equal svamd_cross_correlate_1d with a workspace, bit for bit.'''
    import _native as nv
    torch = nv.torch_cuda()
    _warm(nv, torch)
    rng = np.random.default_rng(9)
    n = 3000
    d1, d2 = nv.to_device(rng.random(n)), nv.to_device(rng.random(n))
    got = nv.empty(2 * n - 1, torch.float64)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        nv.call('svamd_cross_correlate_1d_static', nv.ptr(d1), nv.ptr(d2), n, 1, nv.ptr(got), nv.stream_ptr())
    ws = nv.workspace(nv.load_library().svamd_cross_correlate_1d_workspace_bytes(n))
    want = nv.empty(2 * n - 1, torch.float64)
    for seed in (1, 2):
        fresh = np.random.default_rng(seed)
        d1.copy_(torch.from_numpy(fresh.random(n)))
        d2.copy_(torch.from_numpy(fresh.standard_normal(n)))
        got.fill_(float('nan'))
        graph.replay()
        nv.call('svamd_cross_correlate_1d', nv.ptr(d1), nv.ptr(d2), n, 1, nv.ptr(want), nv.ptr(ws), ws.numel(),
                nv.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(got, want)
