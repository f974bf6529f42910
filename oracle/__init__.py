"""TEST INFRASTRUCTURE ONLY -- the parity checker of parcels_amd, never part of the product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import from here:

* ``ref_shim``        loads the reference's own hot-path modules from /root/reference under a stub of xarray et al.
* ``make_golden``     runs the reference on the neutral cases of ``cases`` and writes ``tests/golden/*.npz``
* ``make_v3_golden``  decodes the reference's v3-JIT regression data (``mini_hdf5``, ``mini_zarr``) into fixtures
* ``parcels_oracle.c`` / ``c_oracle``  scalar C restatement of the path (OpenMP), pinned bit-for-bit to the fixtures
"""
