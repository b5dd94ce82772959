"""Host-side mirror of ``skoots.lib`` for the eval hot path (same names and argument
meaning as the reference; torch ROCm tensors in, HIP kernels underneath)."""
