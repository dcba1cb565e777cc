"""Parts of bench.py (repo root): workloads and synthetic operands, the multi-rank launcher and its CPU dry run, HIP events,
the roofline arithmetic, the CPU baseline and the timed regions of one workload.  bench.py itself only parses arguments,
runs the headline workload (plus one decoder layer of every other BASELINE configuration on the default line) and prints
the driver's ONE JSON line.  Nothing here is imported by the product (lqer_amd/)."""
