"""The evaluation glue (hashgan_amd/evalio.py) against the reference's own data files where they
exist (build container), and on synthetic inputs everywhere."""
import os
import numpy as np
import pytest
from hashgan_amd import evalio

REF = "/root/reference"


def test_read_label_list_and_stack(tmp_path):
    p = tmp_path / "list.txt"
    p.write_text("a/0.jpg 0 1 0\nb/1.jpg 1 0 0\n\nc/2.jpg 0 0 1\n")
    lab, paths = evalio.read_label_list(str(p), with_paths=True)
    assert lab.dtype == np.int64 and lab.tolist() == [[0, 1, 0], [1, 0, 0], [0, 0, 1]]
    assert paths == ["a/0.jpg", "b/1.jpg", "c/2.jpg"]
    # two batches of 2, the last padded by wrap-around; size=3 drops the padding (main.py:157-158)
    outs = [np.arange(8, dtype=np.float32).reshape(2, 4), np.arange(8, 16, dtype=np.float32).reshape(2, 4)]
    labs = [lab[:2], np.stack([lab[2], lab[0]])]
    s = evalio.stack_batches(outs, labs, 3, 4, 3)
    assert s.output.shape == (3, 4) and s.label.tolist() == lab.tolist()
    assert s.output[2].tolist() == [8, 9, 10, 11]


def test_config_defaults_and_yaml(tmp_path):
    c = evalio.load_eval_config()
    assert (c.HASH_DIM, c.LABEL_DIM, c.DB_SIZE, c.TEST_SIZE, c.MAP_R) == (64, 10, 54000, 1000, 54000)
    y = tmp_path / "c.yaml"
    y.write_text("DATA:\n  LABEL_DIM: 81\n  DB_SIZE: 168692\n  TEST_SIZE: 5000\n  MAP_R: 5000\n")
    c = evalio.load_eval_config(str(y))
    assert (c.HASH_DIM, c.LABEL_DIM, c.DB_SIZE, c.TEST_SIZE, c.MAP_R) == (64, 81, 168692, 5000, 5000)


@pytest.mark.reference
@pytest.mark.skipif(not os.path.exists(REF + "/config/cifar_evaluation.yaml"), reason="no /root/reference")
def test_against_reference_files():
    c = evalio.load_eval_config(REF + "/config/cifar_evaluation.yaml")
    assert (c.LABEL_DIM, c.DB_SIZE, c.TEST_SIZE, c.MAP_R) == (10, 54000, 1000, 54000)
    lab = evalio.read_label_list(REF + "/data_list/cifar10/test.txt")
    assert lab.shape == (c.TEST_SIZE, c.LABEL_DIM) and (lab.sum(1) == 1).all()
    from tests import cases
    z = np.load(os.path.join(cases.GOLDEN_DIR, "cifar10_labels.npz"))
    assert np.array_equal(lab.argmax(1), z["test_cls"])           # the committed fixture is this file
    c = evalio.load_eval_config(REF + "/config/nuswide_step_1.yaml")
    assert c.LABEL_DIM == 81 and c.MAP_R == 5000


@pytest.mark.gpu
def test_evaluate_like_main_py(case_cache):
    from tests import cases
    c = case_cache("e_b100")
    g = cases.load_golden("e_b100")
    db = evalio.stack_batches([np.tanh((c["dbbits"].astype(np.float32) * 2 - 1) * 2)], [c["dblab"]],
                              c["dbbits"].shape[0], c["b"], c["dblab"].shape[1])
    test = evalio.stack_batches([np.tanh((c["qbits"].astype(np.float32) * 2 - 1) * 2)], [c["qlab"]],
                                c["qbits"].shape[0], c["b"], c["qlab"].shape[1])
    # the hashing evaluation proper: sign() first, Hamming ranking -> the golden of the +-1 codes
    assert evalio.report(evalio.evaluate(db, test, c["R"], binarize=True)) == g["map"]
    # the default ranks the raw tanh features by inner product, like main.py:164 does -- a different number
    from hashgan_amd import MAPs
    m = MAPs(c["R"])
    raw = m.get_maps_by_feature(db, test)
    m.close()
    assert evalio.evaluate(db, test, c["R"]) == raw != g["map"]
