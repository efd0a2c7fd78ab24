"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference hot path (AsteriosPar/mmWave_MSc:
src/Tracking.py, src/Utils.py, src/constants.py, src/train.py) used as the
parity checker.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import anything below this package; the product
(`mmwave_msc_amd`) never does.
"""
