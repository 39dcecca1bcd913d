"""LDS read-rate microbenchmark (developer tool, trace build): clocks per wave-level read instruction with
4 waves of one workgroup reading concurrently, for the pixel strides considered for the LDS tiles."""
import ctypes as C, os
import torch
L = C.CDLL(os.environ['SISR_LIB'])
out = torch.zeros(8, dtype=torch.int64, device='cuda')
names = {0: 'ds_read_b64_tr_b16 (wgrad pattern)', 1: 'ds_read_b64 (same addresses)', 2: 'ds_read_b128 (lane = pixel)'}
for kind in (0, 1, 2):
    for ps in (32, 40, 48, 72, 96, 160):
        if (31 + 12) * ps + 64 > 40 * 1024:
            continue
        iters = 2000
        rc = L.sisr_lds_bench(kind, ps, iters, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        clk = out[:4].float().mean().item() / (iters * 12)
        print('%-36s stride %3d elements: %.1f clocks per read per wave (4 waves concurrently)' % (names[kind], ps, clk))
