"""Light -- mirror of the reference's `interface Light` (src/rendering-raycast/light.ts:3-7)."""
from dataclasses import dataclass, field
from typing import List


@dataclass
class Light:
    # plain JS numbers in the reference (f64 until packed into the Float32Array, RR:163-164)
    position: List[float] = field(default_factory=lambda: [0.0, 5.0, 0.0])
    lightIntensity: float = 3.0
    minIntensity: float = 0.3
