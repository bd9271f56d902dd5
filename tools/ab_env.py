"""A/B of the config-2 step time under an environment switch read by the kernel library (each arm in its own process,
arms interleaved twice on the same box).  usage: ab_env.py VAR"""
import os, subprocess, sys
var = sys.argv[1]
code = ("import sys,os,torch; sys.path.insert(0,'.'); import bench; dev=torch.device('cuda',0); b,_=bench.synth(0,dev); "
        "m=bench.make_model('dense',dev); el,_=bench.timed_steps(m,b,300,20,1,None,dev); print('%.4f ms/step' % (el/300*1e3))")
for rep in range(2):
    for on in (False, True):
        env = dict(os.environ)
        if on: env[var] = "1"
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print(f"{var}={'1' if on else '-'}: {out}")
