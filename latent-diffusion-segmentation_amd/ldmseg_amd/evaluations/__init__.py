from .panoptic_evaluation_agnostic import PanopticEvaluatorAgnostic, pq_compute, id2rgb, rgb2id, get_table  # noqa: F401
