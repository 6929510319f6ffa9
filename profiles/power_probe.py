#!/usr/bin/env python3
'''Samples rocm-smi clocks / temperatures while the band kernel runs back to back
(4096x4096, 31x31, 257 offsets): shows the engine clock the device sustains under
this load (2.14-2.21 GHz on the boxes measured, against the 2.4 GHz the FP32
peak is quoted at).'''
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'stereo-vision_amd'))
import torch          # noqa: E402
import bench          # noqa: E402
import band_scan as bs  # noqa: E402
H = W = 4096
l, r = bench.synthetic_pair(H, W, 1)
sc = bs.BandScanner(H, W, 31, 31, 128)
sc.upload(l, r); sc.prepare()
stop = False
def load():
    while not stop:
        for _ in range(10):
            sc.run(0.5, prepared=True)
        torch.cuda.synchronize()
t = threading.Thread(target=load); t.start()
time.sleep(1.0)
for i in range(4):
    out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp'], capture_output=True, text=True).stdout
    print('\n'.join(x for x in out.split('\n') if any(k in x for k in ('Power', 'sclk', 'mclk', 'Temp', 'fclk'))))
    print('---', flush=True)
    time.sleep(0.7)
stop = True; t.join()
out = subprocess.run(['rocm-smi', '--showmaxpower', '--showclkfrq'], capture_output=True, text=True).stdout
print(out[:3000])
