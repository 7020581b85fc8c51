'use strict';
// Node -- src/rendering-raycast/acceleration/node.ts:3-8
class Node { constructor() { this.minCorner = null; this.leftChildIndex = 0; this.maxCorner = null; this.primitiveCount = 0; } }
module.exports = { Node };
