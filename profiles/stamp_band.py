#!/usr/bin/env python3
'''Progress curves of the band kernel's wavefronts, from a diagnostic build of the
library (-DSVAMD_BAND_STAMPS): for a few workgroups, when each wavefront finished
each 8-offset pass.  Usage: python3 profiles/stamp_band.py <libsvamd_stamps.so>'''
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'profiles'))
import bench       # noqa: E402
import ab_band     # noqa: E402

H = W = 4096
lib = ab_band.load(sys.argv[1])
left, right = bench.synthetic_pair(H, W, 1000)
dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
nwy = nwx = H - 30
disp = torch.empty((nwy, nwx), dtype=torch.int32, device='cuda')
val = torch.empty((nwy, nwx), dtype=torch.float64, device='cuda')
ws = torch.empty(lib.svamd_band_scan_u8_workspace_bytes(H, W, 31, 31), dtype=torch.uint8, device='cuda')
nblocks = 16 * 255
stamps = torch.zeros(nblocks * 16 * 40, dtype=torch.int64, device='cuda')
for _ in range(2):
    rc = lib.svamd_band_scan_u8(dl.data_ptr(), dr.data_ptr(), H, W, 31, 31, 1, 1, 128, 1, 0.5, 0.01, 0, nwy, 0,
                                disp.data_ptr(), val.data_ptr(), stamps.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == 0
    torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(nblocks, 16, 40)
np.set_printoptions(linewidth=250, suppress=True)
ends = (s[:, :, 32] - s[:, :, 0].min(axis=1, keepdims=True)) / 1e5
full = np.where(ends.max(1) > 0.9 * np.median(ends.max(1)))[0]
print('block duration ms: median %.3f' % np.median(ends[full].max(1)))
print('kernel span ms: %.3f' % ((s[:, :, 32].max() - s[:, :, 0].min()) / 1e5))
for blk in full[[1000, 2500]]:
    b = s[blk]
    t0 = b[:, 0].min()
    print('block', blk)
    for w in (0, 4, 8, 12):
        t = (b[w, :33] - t0) / 1e5
        print('  wave %2d pass end (ms):' % w, np.round(t[[1, 2, 4, 8, 12, 16, 20, 24, 28, 32]], 3))

# per-block accounting over all full blocks: where a wavefront's share of the block time goes
b = s[full].astype(np.float64)
t0 = b[:, :, 0].min(axis=1, keepdims=True)
first, last = b[:, :, 1] - t0, b[:, :, 32] - t0          # end of pass 1 / of pass 32, per wavefront
block = last.max(axis=1, keepdims=True)
per_pass = (last - first) / 31.0
lead_in = first - per_pass                                # entry -> start of the first 8-offset pass
print('per block (mean over blocks and wavefronts), share of the block duration:')
print('  staging + centre pass before the first pass : %.3f' % np.mean(lead_in / block))
print('  32 passes                                   : %.3f' % np.mean((last - lead_in) / block))
print('  waiting for the slowest wavefront at the end : %.3f' % np.mean((block - last) / block))
print('  spread of pass time between wavefronts (max/min per block): %.3f' % np.mean(per_pass.max(1) / per_pass.min(1)))
for blk in full[[1000, 2500]]:
    bb = s[blk]
    print('block', blk, 'end of pass 32 per wavefront (ms):', np.round((bb[:, 32] - bb[:, 0].min()) / 1e5, 2))
    print('          end of pass 8  per wavefront (ms):', np.round((bb[:, 8] - bb[:, 0].min()) / 1e5, 2))
