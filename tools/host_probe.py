#!/usr/bin/env python3
"""Host-side launch cost of the box next to the GPU: empty-ish kernel launches per second from one thread, load average, CPU clock.
The single-call legs of bench.py are a dozen dependent launches of 30-1000 us each: a host that needs 100 us per launch shows up as GPU idle time."""
import os, time, torch
x = torch.zeros(64, device='cuda')
for _ in range(200): x.add_(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): x.add_(1)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
mhz = [float(l.split(':')[1]) for l in open('/proc/cpuinfo') if l.startswith('cpu MHz')]
print('host_probe: %.1f us per launch (submit), %.1f us per launch (submit + drain), loadavg %s, cpus %d, cpu MHz min/avg/max %.0f/%.0f/%.0f' % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6, open('/proc/loadavg').read().split()[:3], os.cpu_count(), min(mhz), sum(mhz) / len(mhz), max(mhz)))
