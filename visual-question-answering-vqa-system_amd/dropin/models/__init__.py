"""Drop-in replacement for the reference's `models` package: put this directory's parent on sys.path ahead of the
reference checkout and `from models.vqa_model import VQAModel, create_vqa_model, load_vqa_model` resolves here."""
