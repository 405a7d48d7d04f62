"""The end-to-end evaluation legs alone (bench.end_to_end_leg: tuned call and the default call) with their phase times:
python tools/prof_startup.py"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
os.environ['FSVIT_CHUNK'] = '12800'
r = bench.end_to_end_leg(argparse.Namespace(episodes=128), torch.device('cuda', 0))
d = r.pop('default_call')
for k in ('value', 'loop_seconds', 'evaluate_call_seconds', 'phase_seconds', 'episodes_per_launch'):
    print('tuned  ', k, r[k])
    print('default', k, d[k])
