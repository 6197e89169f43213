"""kgwas_amd -- MI355X-native hot path of snap-stanford/KGWAS behind the reference's Python API.

    from kgwas_amd import KGWAS, KGWAS_Data          # same surface as `from kgwas import ...`

The compute path is libkgwas_hip.so (hand-written HIP for gfx950, C ABI in include/kgwas_hip.h);
there is no CPU fallback -- importing is fine anywhere, running needs a ROCm device.
"""
from .graph import HeteroGraph, GraphSchema
from .kgwas_data import KGWAS_Data
from .kgwas import KGWAS
from .model import HeteroGNN, RelationPack, SimpleMLP
from .sampler import NeighborLoader, DeviceGraph, SampledBatch

__version__ = '0.1.0'
__all__ = ['KGWAS', 'KGWAS_Data', 'HeteroGNN', 'RelationPack', 'SimpleMLP', 'NeighborLoader',
           'DeviceGraph', 'SampledBatch', 'HeteroGraph', 'GraphSchema']
