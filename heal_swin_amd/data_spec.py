"""Shape contract consumed by the model constructor -- mirrors the reference's `DataSpec`
(heal_swin/data/segmentation/data_spec.py:5-11); only dim_in, f_in, f_out, base_pix are read by the model."""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple, Union


@dataclass
class DataSpec:
    dim_in: Union[int, Tuple[int, int]]  # single int for healpy: number of pixels = base_pix * nside^2
    f_in: int
    f_out: int
    base_pix: Optional[int]
    class_names: List[str] = field(default_factory=list)
