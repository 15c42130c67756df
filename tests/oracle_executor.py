"""The DAG-level walker over the CPU oracle lives with the oracle (oracle/executor.py, r6); the tests keep this name."""
from oracle.executor import Cipher, OracleExecutor, Plain, c_walk, lower  # noqa: F401
