"""taxor_amd.numa: a rank finds its GPU's NUMA node and CPUs from sysfs alone (before any HIP call)."""
import os

from taxor_amd import numa


def _fake_sysfs(tmp_path, gpus):
    """gpus: list of (domain, bus, dev, fn, numa_node, cpulist); node 0 is a CPU node like on real hosts"""
    kfd = tmp_path / "kfd"
    pci = tmp_path / "pci"
    (kfd / "0").mkdir(parents=True)
    (kfd / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (dom, bus, dev, fn, node, cpus) in enumerate(gpus, start=1):
        (kfd / str(i)).mkdir()
        loc = (bus << 8) | (dev << 3) | fn
        (kfd / str(i) / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {loc}\ndomain {dom}\nname gfx950\n")
        d = pci / ("%04x:%02x:%02x.%x" % (dom, bus, dev, fn))
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
        (d / "local_cpulist").write_text(cpus + "\n")
    return str(kfd), str(pci)


def test_parse_cpulist():
    assert numa.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert numa.parse_cpulist("") == set()


def test_gpu_locality_from_sysfs(tmp_path):
    kfd, pci = _fake_sysfs(tmp_path, [(0, 0x05, 0, 0, 0, "0-63,128-191"), (0, 0x85, 0, 0, 1, "64-127,192-255"),
                                       (1, 0x0a, 0, 0, -1, "0-255")])
    assert numa.kfd_gpus(kfd) == ["0000:05:00.0", "0000:85:00.0", "0001:0a:00.0"]
    node, cpus = numa.gpu_locality(1, kfd, pci, env={})
    assert node == 1 and cpus == set(range(64, 128)) | set(range(192, 256))
    assert numa.gpu_locality(2, kfd, pci, env={}) == (None, None)          # numa_node -1: the platform does not say
    assert numa.gpu_locality(7, kfd, pci, env={}) == (None, None)
    # HIP_VISIBLE_DEVICES reorders ordinals
    node, _ = numa.gpu_locality(0, kfd, pci, env={"HIP_VISIBLE_DEVICES": "1,0"})
    assert node == 1
    assert numa.visible_ordinals(3, {"ROCR_VISIBLE_DEVICES": "2,1", "HIP_VISIBLE_DEVICES": "1"}) == [1]
    assert numa.visible_ordinals(3, {"HIP_VISIBLE_DEVICES": "GPU-abc"}) == [0, 1, 2]


def test_bind_to_gpu_sets_affinity(tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    half = allowed[: max(1, len(allowed) // 2)]
    kfd, pci = _fake_sysfs(tmp_path, [(0, 5, 0, 0, 0, ",".join(map(str, half)))])
    try:
        info = numa.bind_to_gpu(0, kfd_nodes=kfd, pci_devices=pci, env={})
        assert info == {"bound": True, "numa_node": 0, "cpus": len(half)}
        assert os.sched_getaffinity(0) == set(half)
    finally:
        os.sched_setaffinity(0, allowed)
    info = numa.bind_to_gpu(3, kfd_nodes=kfd, pci_devices=pci, env={})
    assert info["bound"] is False and os.sched_getaffinity(0) == set(allowed)
