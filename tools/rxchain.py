"""Times the receive chain on the reference's grid (LS estimate with nearest-neighbour
interpolation, then the equalizer object; 4 rx x 2 streams, 8192 slots):
    CET_LIB=... python tools/rxchain.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import channelestimationtransformer_amd as cet

rg = cet.ResourceGrid(num_ofdm_symbols=14, fft_size=76, subcarrier_spacing=15e3, num_tx=1,
                      num_streams_per_tx=2, cyclic_prefix_length=6, num_guard_carriers=[5, 6],
                      dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
est = cet.LSChannelEstimator(rg, interpolation_type="nn")
equ = cet.LMMSEEqualizer(rg, cet.StreamManagement(np.array([[1]]), 2))
nb = 8192
# four input sets used in turn (4 GB in all): nothing a call reads is left in the 256 MB
# memory-side cache by the call before it
SETS = 4
ys = [torch.view_as_complex(torch.randn(nb, 1, 4, 14, 76, 2, device="cuda")) for _ in range(SETS)]
est_out = [est([y, 0.1]) for y in ys]
turn = {"ls": 0, "eq": 0}


def timed(step, n=60):
    for _ in range(20):
        step()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        step()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def step_ls():
    turn["ls"] = (turn["ls"] + 1) % SETS
    est([ys[turn["ls"]], 0.1])


def step_eq():
    k = turn["eq"] = (turn["eq"] + 1) % SETS
    equ([ys[k], est_out[k][0], est_out[k][1], 0.1])


ls = timed(step_ls)
eq = timed(step_eq)
n_out = nb * 4 * 2 * 14 * 64
# bytes of the LS estimate: 12 per output element + the two pilot symbols of every antenna plane
ls_bytes = n_out * 12 + nb * 4 * 2 * 76 * 8
print(f"LS estimate   {ls:.4f} ms  {ls_bytes / ls / 1e9:.2f} TB/s", flush=True)
print(f"equalizer     {eq:.4f} ms  {nb * 12 * 64 * 152.0 / eq / 1e9:.2f} TB/s  {nb * 12 * 64 / eq / 1e6:.2f} G RE/s", flush=True)
