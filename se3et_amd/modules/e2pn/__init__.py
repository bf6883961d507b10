from .blocks_epn import (KPConvInterSO3, GroupNormEPN, UnaryBlockEPN, LastUnaryBlockEPN, KPConvInterSO3Block, SimpleBlockEPN,
                         ResnetBottleneckBlockEPN, InvOutBlockEPN, LiftBlockEPN)
