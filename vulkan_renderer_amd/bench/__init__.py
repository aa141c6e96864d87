"""The pieces of bench.py (the benchmark of the shading pass at the repository root, which the build driver runs):

  common    constants of the roofline, identity of the kernel sources, host probes
  launch    ranks: torch.distributed plumbing, self-launch of `--gpus N`, the rendezvous dry run
  workload  one BASELINE configuration set up, timed and described (run_workload), the other arithmetic modes beside it
  roofline  the `roofline` object: nominal HBM figures, HBM traffic measured in the run, what binds the kernel
  parity    the GPU frames against the CPU oracle, and the oracle timed as the CPU baseline
  line      the ONE short JSON line and the details file

Test infrastructure, not product: nothing here is loaded by libvkr_shading.so or its Python mirror."""
