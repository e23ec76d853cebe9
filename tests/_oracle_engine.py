"""Re-export of the oracle-backed stand-in engine (oracle/engine_standin.py) under its historical test-module name."""
from oracle.engine_standin import OracleNesEngine  # noqa: F401
