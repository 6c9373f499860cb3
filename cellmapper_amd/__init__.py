"""cellmapper_amd: MI355X-native backend for CellMapper's k-NN -> graph kernel -> aggregation hot path.

Drop-in surface (same names as quadbio/cellmapper): ``CellMapper``, ``Neighbors``, ``NeighborsResults``, ``logger``.
"""
from ._anndata import AnnData, MiniAnnData
from .cellmapper import CellMapper
from .knn import Neighbors, NeighborsResults
from .logging import logger

__all__ = ["logger", "CellMapper", "Neighbors", "NeighborsResults", "AnnData", "MiniAnnData"]
__version__ = "0.1.0"
