from typing import Optional, Tuple, Union
from torch import Tensor
Adj = Union[Tensor, object]
OptTensor = Optional[Tensor]
PairTensor = Tuple[Tensor, Tensor]
OptPairTensor = Tuple[Tensor, Optional[Tensor]]
Size = Optional[Tuple[int, int]]
NoneType = Optional[Tensor]
