"""The issue schedule of one k tile of the plane GEMM (k_gemm3p.hip: P3Wave::tile / rows / mfmas / behind), replayed on the host.

The kernel's operand registers are re-used inside a k tile (NAF activation fragment buffers for MI fragment rows; fragment r + 2 is read into
the buffer row r - 1 has released) and its plane reads are issued a fixed number of matrix instructions ahead of their first use.  This test
transcribes the compile-time schedule (which read is issued behind which matrix instruction) and checks, for every tile shape (three bf16
planes, six products), that every matrix instruction of fragment row r, column n, product p finds (a) the weight plane WP[p] of column n and (b) the activation plane
AP[p] of fragment row r in its registers, read EARLIER in the stream; that every product of every (row, column) is issued exactly once; and that
the next tile's DMA instructions are all issued, in order, exactly once."""
import pytest

SHAPES = [(4, 5), (4, 4), (2, 5), (2, 4), (2, 2)]      # (MI, NI) of the instantiated wave tiles
PRODUCTS = {3: ([2, 0, 1, 1, 0, 0], [0, 2, 1, 0, 1, 0])}   # weight plane, activation plane of product p (k_gemm3p.hip mfmas())


def replay(npl, mi, ni, nag, nbw):
    wp, ap = PRODUCTS[npl]
    nprod = len(wp)
    naf = 3 if mi > 2 else mi
    nmf = nprod * ni
    np_ = npl * nag + nbw
    dma_slots = 3 * ni + nmf
    wf = {}                      # (plane, n) -> True once read
    af = {}                      # buffer index -> {plane: fragment row it holds}
    dma = []
    done = set()

    def read_a(f, pl):
        af.setdefault(f % naf, {})[pl] = f

    def read_w(pl, n):
        wf[(pl, n)] = True

    def pieces(j0, j1):
        dma.extend(range(j0, j1))

    def behind(midx, k):
        pr, n = divmod(k, ni)
        if True:
            if midx == 0:
                if pr == 0:
                    read_w(0, n)
                    if n == ni - 1:
                        read_a(0, 2)
                if pr == 1:
                    read_w(1, n)
                    if n == ni - 1:
                        read_a(0, 1)
                if 2 * ni <= k < 2 * ni + 3:
                    read_a(1, k - 2 * ni)
                if mi > 2 and 2 * ni + 3 <= k < 2 * ni + 6:
                    read_a(2, k - 2 * ni - 3)
            elif midx + 2 < mi:
                if k < 3:
                    read_a(midx + 2, k)
            if midx == 0 and pr >= 3:
                slot = k - 3 * ni
                pieces(slot * np_ // dma_slots, (slot + 1) * np_ // dma_slots)
            elif midx == 1:
                slot = 3 * ni + k
                pieces(slot * np_ // dma_slots, (slot + 1) * np_ // dma_slots)

    # tile(): the h plane of fragment 0 and the l (last) weight planes, then the rows
    read_a(0, 0)
    for n in range(ni):
        read_w(npl - 1, n)
    for midx in range(mi):
        for k in range(nmf):
            pr, n = divmod(k, ni)
            assert wf.get((wp[pr], n)), f"row {midx} product {pr} column {n}: weight plane {wp[pr]} not read yet"
            held = af.get(midx % naf, {}).get(ap[pr])
            assert held == midx, f"row {midx} product {pr} column {n}: activation buffer holds plane {ap[pr]} of fragment {held}"
            assert (midx, n, pr) not in done
            done.add((midx, n, pr))
            behind(midx, k)
    assert len(done) == mi * ni * nprod
    assert dma == list(range(np_)), f"DMA instructions issued {dma}, expected 0..{np_ - 1}"


@pytest.mark.parametrize("npl", [3])
@pytest.mark.parametrize("mi,ni", SHAPES)
def test_every_matrix_instruction_finds_its_operands(npl, mi, ni):
    for nwv, wm, wn in ((8, 4, 2), (8, 2, 4), (4, 2, 2), (4, 1, 4), (4, 4, 1)):
        bm, bn = 16 * mi * wm, 16 * ni * wn
        if bm % (16 * nwv):
            continue
        nag = bm // (16 * nwv)
        nbw = ((bn // 16) * npl + nwv - 1) // nwv
        replay(npl, mi, ni, nag, nbw)


def test_a_wrong_schedule_is_caught():
    """the replay is not vacuous: products that want the m plane of a fragment before it is read must trip it"""
    global PRODUCTS
    saved = dict(PRODUCTS)
    try:
        PRODUCTS = {3: ([2, 0, 1, 1, 0, 0], [1, 2, 1, 0, 1, 0])}      # product 0 would multiply by the fragment's m plane, read behind product 1
        with pytest.raises(AssertionError):
            replay(3, 4, 5, 2, 3)
    finally:
        PRODUCTS = saved
