import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
  # the CPU oracle runs on torch's CPU thread pool: size it to the cores this process may use (cgroup quota)
  import torch
  import gill_amd
  from gill_amd.synth import host_cores
  torch.set_num_threads(host_cores())
  gill_amd.configure_hip_runtime()   # the GPU tests replay captured graphs the way bench.py does (explicit opt-in since round 5)


@pytest.fixture(scope="session")
def cuda():
  import torch
  if not torch.cuda.is_available():
    pytest.skip("no GPU visible")
  return torch.device("cuda:0")
