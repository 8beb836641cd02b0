"""The Python-visible call shape of the hot path (last row): what `pyhdk` offers for it --
`hdk.import_arrow(table, name, fragment_size)` / `import_pydict`, then run one execution step on a device and
get a `pyarrow.Table` (python/pyhdk/hdk.py:2361, python/pyhdk/_sql.pyx:169-213 `RelAlgExecutor.execute(
device_type=...)` -> `ExecutionResult.to_arrow()`).  The step is given as a `QueryUnit` (the stand-in for the
RelAlgExecutionUnit that HDK's planner would hand to the kernels); SQL parsing and RelAlg planning stay above
this boundary.  device_type="CPU" raises QueryMustRunOnCpu: there is no CPU path in this package."""
from typing import Dict, Optional, Sequence

import numpy as np

from .executor import Executor
from .hip_mgr import HipMgr
from .ir import QueryMustRunOnCpu, QueryUnit
from .storage import ArrowStorage


class Engine:
    def __init__(self, device_id: int = 0, mgr: Optional[HipMgr] = None):
        self.storage = ArrowStorage()
        self.mgr = mgr or HipMgr()
        self.device_id = device_id
        self._executor: Optional[Executor] = None

    # ---- import (pyhdk: HDK.import_arrow / import_pydict, hdk.py:2361-2420) ------------------------------
    def import_arrow(self, table, name: str, fragment_size: Optional[int] = None):
        self._executor = None
        return self.storage.import_arrow(table, name, fragment_size=fragment_size)

    def import_pydict(self, values: Dict[str, Sequence], name: str, fragment_size: Optional[int] = None):
        import pyarrow as pa
        return self.import_arrow(pa.table({k: pa.array(v) for k, v in values.items()}), name, fragment_size)

    def import_numpy(self, name: str, columns: Dict[str, np.ndarray], fragment_size: Optional[int] = None, **kw):
        self._executor = None
        return self.storage.import_numpy(name, columns, fragment_size=fragment_size, **kw)

    # ---- run one step ----------------------------------------------------------------------------------------
    def executor(self) -> Executor:
        if self._executor is None:
            self._executor = Executor(self.storage, self.device_id, self.mgr)
        return self._executor

    def execute(self, q: QueryUnit, device_type: str = "GPU", **kw):
        """-> ExecutionResult (buffer in the reference's layout, error code, to_arrow())."""
        if device_type.upper() != "GPU":
            raise QueryMustRunOnCpu("this package ships the GPU path only")
        return self.executor().execute(q, **kw)

    def run(self, q: QueryUnit, device_type: str = "GPU", **kw):
        """-> pyarrow.Table with the targets' names and types, rows in buffer order."""
        return self.execute(q, device_type, **kw).to_arrow()
