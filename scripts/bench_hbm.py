"""Achievable HBM bandwidth of the box for plain streams (torch elementwise kernels on fp64), the yardstick next
to the 8 TB/s peak: read-only sum, copy (1 read + 1 write), triad-like a = b + s * c (2 reads + 1 write)."""
import sys
import torch


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    for mb in (64, 512, 2048):
        n = mb * (1 << 20) // 8
        a = torch.zeros(n, dtype=torch.float64, device="cuda")
        b = torch.ones(n, dtype=torch.float64, device="cuda")
        c = torch.ones(n, dtype=torch.float64, device="cuda")
        t_copy = timed(lambda: a.copy_(b))
        t_triad = timed(lambda: torch.add(b, c, alpha=2.0, out=a))
        t_read = timed(lambda: torch.sum(b))
        print(f"{mb:5d} MB per array: read {n * 8 / t_read / 1e12:5.2f} TB/s   copy {2 * n * 8 / t_copy / 1e12:5.2f} TB/s   "
              f"triad {3 * n * 8 / t_triad / 1e12:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    sys.exit(main())
