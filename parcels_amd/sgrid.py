"""Minimal SGRID metadata model (the part of src/parcels/_sgrid/core.py the hot path reads)."""

from __future__ import annotations

import enum
from dataclasses import dataclass


class Padding(enum.Enum):  # _sgrid/core.py:34-38
    NONE = "none"
    LOW = "low"
    HIGH = "high"
    BOTH = "both"


def get_n_faces(n_nodes: int, padding: Padding) -> int:  # _sgrid/core.py:41-49
    if padding in (Padding.LOW, Padding.HIGH):
        return n_nodes
    if padding == Padding.NONE:
        return n_nodes - 1
    if padding == Padding.BOTH:
        return n_nodes + 1
    raise ValueError(f"Invalid {padding=!r}")


@dataclass
class FaceNodePadding:  # _sgrid/core.py:331-362
    face: str
    node: str
    padding: Padding


class SGrid2DMetadata:  # _sgrid/core.py:70-190
    def __init__(self, cf_role="grid_topology", topology_dimension=2, node_dimensions=None, face_dimensions=None,
                 node_coordinates=None, vertical_dimensions=None):
        if cf_role != "grid_topology":
            raise ValueError(f"cf_role must be 'grid_topology', got {cf_role!r}")
        if topology_dimension != 2:
            raise ValueError("topology_dimension must be 2")
        if node_dimensions is None or len(node_dimensions) != 2:
            raise ValueError("node_dimensions must be a pair of dimension names")
        if face_dimensions is None or len(face_dimensions) != 2:
            raise ValueError("face_dimensions must be a pair of FaceNodePadding")
        self.cf_role = cf_role
        self.topology_dimension = topology_dimension
        self.node_dimensions = tuple(node_dimensions)
        self.face_dimensions = tuple(face_dimensions)
        self.node_coordinates = tuple(node_coordinates) if node_coordinates is not None else None
        self.vertical_dimensions = tuple(vertical_dimensions) if vertical_dimensions is not None else None

    def dim_to_axis(self) -> dict:  # _sgrid/accessor.py:125-138
        fx, fy = self.face_dimensions
        d = {fx.node: "X", fx.face: "X", fy.node: "Y", fy.face: "Y"}
        if self.vertical_dimensions is not None:
            fz = self.vertical_dimensions[0]
            d.update({fz.node: "Z", fz.face: "Z"})
        return d
