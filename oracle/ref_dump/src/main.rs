//! Writes a small single-field index with the real crate, runs a fixed set of queries through `Search::search` and dumps
//! the answers.  Mirrors the calls of the reference's own tests (tests/test.rs:20-210).  NOT built in this repo's image
//! (no cargo): a recipe for a box that has it; see Cargo.toml.
use seekstorm::commit::Commit;
use seekstorm::index::{
    AccessType, Close, Clustering, DocumentCompression, FrequentwordType, IndexDocuments, IndexMetaObject, LexicalSimilarity,
    NgramSet, StemmerType, StopwordType, TokenizerType, create_index, open_index,
};
use seekstorm::search::{QueryRewriting, QueryType, ResultType, Search, SearchMode};
use seekstorm::vector::Inference;
use std::{env, fs, path::Path};

#[tokio::main]
async fn main() {
    let out = env::args().nth(1).unwrap_or_else(|| "dump".into());
    let index_path = Path::new(&out).join("index_test");
    let _ = fs::remove_dir_all(&index_path);
    fs::create_dir_all(&index_path).unwrap();
    let schema = serde_json::from_str(r#"[{"field":"body","field_type":"Text","store":true,"index_lexical":true,"longest":true}]"#).unwrap();
    let meta = IndexMetaObject {
        id: 0,
        name: "dump".into(),
        lexical_similarity: LexicalSimilarity::Bm25f,
        tokenizer: TokenizerType::UnicodeAlphanumeric,
        stemmer: StemmerType::None,
        stop_words: StopwordType::None,
        frequent_words: FrequentwordType::None,
        ngram_indexing: NgramSet::SingleTerm as u8,
        document_compression: DocumentCompression::Snappy,
        access_type: AccessType::Mmap,
        spelling_correction: None,
        query_completion: None,
        clustering: Clustering::None,
        inference: Inference::None,
    };
    // ONE shard, so that index.bin holds the whole corpus (tests/golden loaders take one shard file)
    let index_arc = create_index(&index_path, meta, &schema, &Vec::new(), 11, false, Some(1)).await.unwrap();
    // a deterministic corpus: doc i holds word w<j> (i % (j + 2) == 0 ? several times : once) for a few j
    let mut docs = Vec::new();
    for i in 0..5000u32 {
        let mut body = String::new();
        for j in 0..12u32 {
            if i % (j + 2) == 0 {
                for _ in 0..(1 + (i / 7) % (j + 1)) {
                    body.push_str(&format!("w{} ", j));
                }
            }
        }
        body.push_str(&format!("pad{} pad{}", i % 97, i % 89));
        docs.push(serde_json::json!({ "body": body }));
    }
    let documents_vec = serde_json::from_value(serde_json::Value::Array(docs)).unwrap();
    index_arc.index_documents(documents_vec).await;
    index_arc.commit().await;
    index_arc.close().await;

    let index_arc = open_index(&index_path).await.unwrap();
    let queries: Vec<(&str, QueryType)> = vec![
        ("w0 w1", QueryType::Union),
        ("w0 w1 w2", QueryType::Union),
        ("+w3 +w5", QueryType::Intersection),
        ("w7", QueryType::Union),
        ("\"w0 w1\"", QueryType::Phrase),
        ("w2 -w3", QueryType::Union),
    ];
    let mut expected = Vec::new();
    for (q, qt) in queries {
        for rt in [ResultType::Topk, ResultType::TopkCount, ResultType::Count] {
            let ro = index_arc
                .search(q.to_string(), None, qt.clone(), SearchMode::Lexical, false, 0, 10, rt.clone(), false, Vec::new(), Vec::new(),
                        Vec::new(), Vec::new(), QueryRewriting::SearchOnly)
                .await;
            expected.push(serde_json::json!({
                "query": q, "query_type": format!("{:?}", qt), "result_type": format!("{:?}", rt),
                "doc_ids": ro.results.iter().map(|r| r.doc_id).collect::<Vec<_>>(),
                "scores": ro.results.iter().map(|r| r.score).collect::<Vec<_>>(),
                "result_count": ro.result_count, "result_count_total": ro.result_count_total,
            }));
        }
    }
    fs::write(Path::new(&out).join("expected.json"), serde_json::to_string_pretty(&expected).unwrap()).unwrap();
    index_arc.close().await;
    println!("wrote {}", out);
}
