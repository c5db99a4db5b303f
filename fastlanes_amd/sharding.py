"""Block-range sharding of a column across the GPUs of a node.

Every 1024-value block is independent in every hot-path function (pack/unpack take
one block, bitpacking.rs:19,33; Delta's bases are per block, delta.rs:7), so a column
shards embarrassingly: rank g owns a contiguous block range and its slice of the packed
bytes; no collective touches the data path."""
import numpy as np


def block_range(n_blocks, world_size, rank):
    """Contiguous range [start, start+count) of rank `rank`; the first n_blocks % world_size
    ranks hold one extra block (10 B ints over 8 GPUs: 9 765 625 blocks -> 1 220 704 on
    rank 0, 1 220 703 on ranks 1..7)."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, rem = divmod(n_blocks, world_size)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def packed_offsets(widths):
    """Byte offset of every block in a mixed-width packed column (exclusive prefix sum of
    128*W; a packed block is 128*W bytes, bitpacking.rs:77), plus the total."""
    w = np.asarray(widths, dtype=np.uint64)
    sizes = w * np.uint64(128)
    off = np.zeros(len(w) + 1, dtype=np.uint64)
    np.cumsum(sizes, out=off[1:])
    return off[:-1], int(off[-1])


def shard_mixed(widths, world_size, rank):
    """(first block, block count, first packed byte, packed byte count) of rank's slice of a
    mixed-width column.  Cuts are block-aligned because offsets are per block."""
    start, count = block_range(len(widths), world_size, rank)
    off, total = packed_offsets(widths)
    b0 = int(off[start]) if start < len(widths) else total
    b1 = int(off[start + count]) if start + count < len(widths) else total
    return start, count, b0, b1 - b0
