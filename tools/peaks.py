"""The two rooflines SURVEY 8(d) asks to re-derive on the box: dense fp16 MFMA peak from CU count x clock, HBM bandwidth from device
copies / a triad / a read-only pass (torch kernels: plumbing, not the product).  python tools/peaks.py"""
import subprocess, time
import torch

dev = torch.device("cuda", 0)
p = torch.cuda.get_device_properties(0)
cus = p.multi_processor_count
mhz = getattr(p, "clock_rate", 0) / 1e3
mem_khz, bus = getattr(p, "memory_clock_rate", 0), getattr(p, "memory_bus_width", 0)
if not mhz:                                   # this torch build does not expose the clocks: ask rocminfo
    import re
    txt = subprocess.run(["rocminfo"], capture_output=True, text=True).stdout
    gpu = txt[txt.find("gfx950"):] if "gfx950" in txt else txt          # the first agent block of the GPU, from its Name line on
    m = re.search(r"Max Clock Freq\. \(MHz\):\s+(\d+)", gpu)
    mhz = float(m.group(1)) if m else 2400.0
print(f"{p.name}: {cus} CUs, max engine clock {mhz:.0f} MHz, {p.total_memory / 2**30:.0f} GiB, memory clock {mem_khz / 1e3:.0f} MHz, bus {bus} bit")
# gfx950: v_mfma_f32_32x32x16_f16 = 32 768 flop in 32 cycles per SIMD, 4 SIMDs per CU
print(f"dense fp16 MFMA peak = {cus} CUs x 4 SIMDs x 1024 flop/cycle x {mhz / 1e3:.2f} GHz = {cus * 4 * 1024 * mhz * 1e6 / 1e15:.2f} PFLOP/s "
      f"(MI355X_MICROARCH.md: ~2.5; measured MFMA-only streams: tools/ubench/mfma_shape_power.hip)")
if bus and mem_khz:
    print(f"HBM pins: {bus} bit x {mem_khz / 1e6:.3f} GHz x 2 (DDR) / 8 = {bus * mem_khz * 1e3 * 2 / 8 / 1e12:.2f} TB/s")
n = 1 << 30                                   # 4 GiB of float32 per array
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
c = torch.empty_like(a)


def timed(fn, bytes_moved, name, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:34s} {bytes_moved / dt / 1e12:5.2f} TB/s  ({dt * 1e3:.2f} ms)")


timed(lambda: b.copy_(a), 2 * 4 * n, "copy  b = a        (read + write)")
timed(lambda: torch.add(a, b, alpha=2.0, out=c), 3 * 4 * n, "triad c = a + 2 b    (2 reads + write)")
timed(lambda: a.sum(), 4 * n, "sum(a)              (read only)")
timed(lambda: b.fill_(1.0), 4 * n, "fill b              (write only)")
