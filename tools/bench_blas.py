"""Library yard-stick for the conv1d GEMM shapes: torch.matmul (hipBLASLt / rocBLAS) fp16 on the same M, N, K.
Not used by the product -- only to know how far the hand-written kernel is from what the vendor library reaches."""
import torch

def run(M, N, K, iters=20):
    a = torch.randn(M, K, device='cuda', dtype=torch.float16)
    b = torch.randn(N, K, device='cuda', dtype=torch.float16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f'M={M} N={N} K={K}: {us:.1f} us  {2.0 * M * N * K / us * 1e-6:.0f} TFLOP/s', flush=True)

if __name__ == '__main__':
    for shape in [(76800, 3072, 3072), (76800, 1024, 1024), (76800, 1024, 3072), (76800, 128, 3072), (76800, 3072, 128),
                  (8192, 8192, 8192)]:
        run(*shape)
