"""bench.py's launch line on the CPU: the same program, argument parsing, sharding, protocol and
JSON line, with the oracle-backed test engine in place of HipEngine (no GPU, gloo).  Started by
tests/test_driver_cpu.py under torch.distributed.run exactly as the driver starts bench.py; the
line it prints says REHEARSAL in `metric` and measures nothing."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import bench                                    # noqa: E402
from oracle_engine import OracleEngine          # noqa: E402

if __name__ == '__main__':
    bench.main(engine_factory=OracleEngine)
