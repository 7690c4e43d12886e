"""The synthetic inputs and graph builders live in the package (phonic_amd/workloads.py: product-side harness code shared by
bench.py, smoke() and the tests); this module keeps the tests' `import workloads` working."""
from phonic_amd.workloads import *  # noqa: F401,F403
from phonic_amd.workloads import reverb_seeds, splitmix64, test_signal, tone_buffer, voice_freq, voice_level, voice_pan  # noqa: F401
