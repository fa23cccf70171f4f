import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from llama_go_amd.mlapi import SHAPES, load_product, make_hparams
prod = load_product()
hp = make_hparams(**SHAPES["7B"], ctx=128)
m = prod.NewSyntheticModel(hp, 1234)
c = m.NewContext(128, 1)
prompt = [1, 15043, 3186, 29892, 445, 338, 263, 1243]
for r in range(4):
    t0 = time.perf_counter(); lg = c.Eval(prompt, 0); t1 = time.perf_counter()
    print(f"python Eval(8) {1e6*(t1-t0):.0f} us", file=sys.stderr)
tok = int(np.argmax(lg))
for i in range(6):
    t0 = time.perf_counter(); lg = c.Eval([tok], 8 + i); t1 = time.perf_counter()
    print(f"python Eval(1) {1e6*(t1-t0):.0f} us", file=sys.stderr)
    tok = int(np.argmax(lg))
