"""Black-box OCR adapters.  Contract (reference ocr_helper/tess_helper.py:10-44):
    helper = Helper(empty_char=' ', is_eval=False)
    helper.get_labels(imgs: float CPU tensor [N,1,H,W] in [0,1]) -> list[str] (len N)
    helper.get_string(img) -> list[str];  helper.count_calls: int
The engines themselves (Tesseract / EasyOCR / Google Vision) are CPU or network black boxes and out
of scope; `stub_helper.StubHelper` implements the same contract deterministically for plumbing runs
and benchmarks."""
