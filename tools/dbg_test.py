import os, sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import tests.test_gpu_parity as T
try:
    T.test_convsbs_classifier_step_eager_and_graphed(4, True)
    print("PASSED as a plain call")
except AssertionError as e:
    print("FAILED as a plain call", str(e)[:300])
